"""BaseExperiment: mirror of models/base.py (seeding, setup order, checkpoint save/resume)."""
import os
import random

import numpy as np
import torch

from ..lib import utils


class BaseExperiment():
    def __init__(self, config, **kwargs):
        self.config = config

    def setup_log(self):
        pass

    def setup_random_seed(self):
        """models/base.py:33-39."""
        torch.manual_seed(self.config['random_seed'])
        if torch.cuda.is_available():
            torch.cuda.manual_seed(self.config['random_seed'])
        np.random.seed(self.config['random_seed'])
        random.seed(self.config['random_seed'])

    def setup_train_data(self):
        pass

    def setup_model(self):
        pass

    def setup_loss(self):
        pass

    def setup_optimizer(self):
        pass

    def setup_train(self):
        """models/base.py:53-59 (same order)."""
        self.setup_log()
        self.setup_random_seed()
        self.setup_model()
        self.setup_loss()
        self.setup_train_data()
        self.setup_optimizer()

    def train(self, **kwargs):
        raise NotImplementedError()

    def train_one_epoch(self, **kwargs):
        raise NotImplementedError()

    def validate(self, **kwargs):
        raise NotImplementedError()

    @staticmethod
    def save_checkpoint(state, is_best, path, prefix=None, name='checkpoint.pth.tar', max_keep=1):
        """models/base.py:70-78."""
        if not os.path.exists(path):
            os.makedirs(path)
        name = '_'.join([prefix, name]) if prefix else name
        best_name = '_'.join([prefix, 'model_best.pth.tar']) if prefix else 'model_best.pth.tar'
        torch.save(state, os.path.join(path, name))
        if is_best:
            torch.save(state, os.path.join(path, best_name))

    initialize_model = staticmethod(utils.initialize_model)   # models/base.py:80-120
