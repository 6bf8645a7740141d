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
        """Logging (tensorboard in the reference) is out of scope for the hot path: nothing to set up."""

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
        """Same files as models/base.py:70-78 writes: <path>/[<prefix>_]<name>, and a second copy [<prefix>_]model_best.pth.tar when
        `is_best` (`max_keep` is accepted and unused there too)."""
        os.makedirs(path, exist_ok=True)
        stem = (prefix + '_') if prefix else ''
        targets = [stem + name] + ([stem + 'model_best.pth.tar'] if is_best else [])
        for fname in targets:
            torch.save(state, os.path.join(path, fname))

    initialize_model = staticmethod(utils.initialize_model)   # models/base.py:80-120
