"""SegmentationExperiment: mirror of models/segmentation.py with the step loop on the HIP path.

Same config keys (train_seg.py:33-61), same setup order, same step semantics (models/segmentation.py:141-157:
train(); zero_grad(); out = model(x); loss = crit(out, y.long()); backward(); Adam.step(); loss.item()), eval =
argmax -> per-class Dice for classes 1..C-1 averaged over volumes then classes (:179-201), validate ->
scheduler.step + checkpoint (:203-239).  Differences, all host-side: the device is explicit
(config['device'], default 'cuda'), data comes from config['data'] == 'synthetic' or injected loaders,
TensorBoard logging is optional, eval Dice is computed on the device (lib/evalMetrics.py), and with
torch.distributed initialised the gradients are averaged with one flat-bucket all-reduce (parallel.py).
"""
import datetime
import os
import time

import numpy as np
import torch
import torch.optim.lr_scheduler as lr_scheduler
from torch.utils.data import DataLoader

from .base import BaseExperiment
from ..lib import datasets as med_data
from ..lib import evalMetrics as metrics
from ..lib.loss import get_loss_function
from ..lib.network_factory import get_network
from ..lib.param_dict import save_dict_to_json
from ..optim import FlatAdam
from .. import ops
from .. import parallel
from .. import trace

try:
    from tensorboardX import SummaryWriter
except Exception:                                             # tensorboardX is optional here
    SummaryWriter = None


class SegmentationExperiment(BaseExperiment):
    def __init__(self, config):
        super(SegmentationExperiment, self).__init__(config)
        self.device = torch.device(self.config.get('device', 'cuda'))
        cfg = self.config
        if cfg['debug_mode']:                                    # debug runs print and validate every second batch / epoch
            print("Debug mode")
            cfg['print_batch_period'] = cfg['valid_epoch_period'] = 2
        self.exp_name = self.experiment_name(cfg)
        # <log_dir>/<experiment name | "debug_seg">/<seed>: the directory layout the reference's checkpoints and resume_dir use
        # (models/segmentation.py:40-42), so runs of either code base can resume each other's checkpoints
        run_dir = "debug_seg" if cfg['debug_mode'] else self.exp_name
        self.ckpoint_dir = os.path.join(cfg['log_dir'], run_dir, str(cfg['random_seed']))
        self.writer = None
        self.global_step = 0
        self.training_data_loader = self.config.get('training_data_loader')
        self.validation_data_loader = self.config.get('validation_data_loader')
        print("Init experiment {} seed {}".format(self.exp_name, self.config['random_seed']))

    @staticmethod
    def experiment_name(cfg):
        """The run's name, field by field as the reference composes it (models/segmentation.py:27-38) -- it is part of the checkpoint path:
        Seg_<model>[_bias][_BN]_<data dir name>_<n>samples_batch_<b>_<e>epochs_<loss>_<weighting>_lr_<lr>[_scheduler_<mode>]."""
        ms = cfg['model_settings']
        parts = ['Seg_', cfg['model']]
        if ms['bias']:
            parts.append('_bias')
        if ms['BN']:
            parts.append('_BN')
        parts += ['_', os.path.basename(cfg['data_dir']),
                  '_%ssamples' % cfg['num_samples'], '_batch_%s' % cfg['batch_size'], '_%sepochs' % cfg['n_epochs'],
                  '_%s_%s' % (cfg['loss'], cfg['loss_settings']['weight_type']), '_lr_%s' % cfg['learning_rate']]
        if cfg['lr_mode'] != 'const':
            parts.append('_scheduler_%s' % cfg['lr_mode'])
        return ''.join(str(v) for v in parts)

    # ---- setup ---------------------------------------------------------------------------------
    def setup_log(self):
        if parallel.rank() != 0:
            return
        if not os.path.isdir(self.ckpoint_dir):
            os.makedirs(self.ckpoint_dir)
        save_dict_to_json(self.config, os.path.join(self.ckpoint_dir, "train_config.json"))
        if SummaryWriter is not None:
            self.writer = SummaryWriter(self.ckpoint_dir)

    def setup_train_data(self):
        if self.training_data_loader is not None:
            return
        dataset = med_data.get_seg_dataset(self.config['data'])
        shape = self.config['synthetic_shape']
        training_data = dataset(self.config["num_samples"] * 2, shape, self.config['n_classes'], seed=self.config['random_seed'])
        sampler = parallel.distributed_sampler(training_data, shuffle=True, seed=self.config['random_seed'])
        self.training_data_loader = DataLoader(training_data, batch_size=self.config['batch_size'], shuffle=sampler is None,
                                               sampler=sampler, num_workers=0)
        validation_data = dataset(self.config.get('num_valid_samples', 2), shape, self.config['n_classes'],
                                  seed=self.config['random_seed'] + 1000)
        self.validation_data_loader = DataLoader(validation_data, batch_size=1, shuffle=False, num_workers=0)

    def setup_model(self):
        model_type = get_network(self.config['model'])
        self.model = model_type(**self.config['model_settings'])
        self.model.to(self.device)
        # softmax-Dice training (train_seg.py:55): the output convolution is evaluated inside the fused head + softmax + Dice kernels
        # (ops.HeadDiceFn) and the logits tensor is only built on demand (train_step's `output`.materialize(), eval mode)
        if (self.config.get('fuse_head_dice', True) and self.config.get('loss') == 'dice'
                and self.config.get('loss_settings', {}).get('softmax') and hasattr(type(self.model), 'lazy_head')):
            self.model.lazy_head = True

    def setup_loss(self):
        self.criterion = get_loss_function(self.config['loss'])(**self.config['loss_settings']).to(self.device)

    def setup_optimizer(self):
        """models/segmentation.py:90-111: Adam + plateau / multiStep / const."""
        self.optimizer = FlatAdam(self.model.parameters(), lr=self.config['learning_rate'])
        # conv weight gradients on a second stream, accumulated into the optimiser's flat bucket (joined in zero_grad / step)
        ops.enable_async_wgrad(bool(self.config.get('async_wgrad', True)))
        # matrix mode of the 3x3x3 convolutions: 'fp32_split' (default, and what bench.py measures: two-term fp16 split per staged tile, 22-bit
        # products on the fp16 matrix pipe -- per product narrower than fp32, bounds in csrc/split_f16.h), 'fp32' (the fp32 matrix instructions:
        # the reference's arithmetic; pass matrix_precision='fp32' for that) or 'bf16' (operands rounded, BASELINE config 5)
        ops.set_matrix_precision(self.config.get('matrix_precision') or ops.DEFAULT_MATRIX_PRECISION)
        self.scheduler = self.make_scheduler(self.optimizer, self.config)

    @staticmethod
    def make_scheduler(optimizer, cfg):
        """Learning-rate schedule of the reference (models/segmentation.py:93-111).  'plateau': x 0.2 when the validation score (maximised)
        has not improved by 0.003 (absolute) for 100 epochs' worth of validations, never below 1e-5.  'multiStep': x gamma at the given
        fractions of n_epochs (the config's `milestones` are replaced by the epoch numbers, as the reference does).  Anything else: constant."""
        mode = cfg['lr_mode']
        if mode == 'plateau':
            validations = 100 // cfg['valid_epoch_period']
            return lr_scheduler.ReduceLROnPlateau(optimizer, mode='max', factor=0.2, patience=validations, threshold=0.003, threshold_mode='abs', min_lr=1e-5)
        if mode == 'multiStep':
            cfg['milestones'] = [int(frac * cfg['n_epochs']) for frac in cfg['milestones']]
            return lr_scheduler.MultiStepLR(optimizer, cfg['milestones'], gamma=cfg['gamma'])
        return None

    # ---- training ------------------------------------------------------------------------------
    def train(self):
        self.setup_train()
        print("Training {}".format(self.exp_name))
        finished_epoch, self.best_score = self.initialize_model(self.model, self.optimizer, self.config['resume_dir'])
        parallel.broadcast_parameters(self.optimizer, model=self.model)
        parallel.pin_host_resources()                 # per-rank core slice + gc.freeze(): the host launch loop is the DP scaling risk
        self.current_epoch = finished_epoch + 1
        for epoch in range(self.current_epoch, self.config['n_epochs'] + 1):
            self.train_one_epoch()
            self.validate()
            self.current_epoch += 1
        if self.writer is not None:
            self.writer.close()
        print('Finished Training: {}'.format(self.exp_name))

    def train_step(self, images, truths):
        """One optimisation step (models/segmentation.py:141-157)."""
        self.model.train()
        with trace.range('seg/zero_grad'):
            self.optimizer.zero_grad()
        with trace.range('seg/forward'):
            output = self.model(images.to(self.device, non_blocking=True))
        with trace.range('seg/loss'):
            loss = self.criterion(output, truths.to(self.device, non_blocking=True))
        with trace.range('seg/backward'):
            loss.backward()
        with trace.range('seg/allreduce'):
            parallel.allreduce_gradients(self.optimizer)
        with trace.range('seg/adam'):
            self.optimizer.step()
        return loss, output

    def train_one_epoch(self):
        running_loss = 0.0
        iters_per_epoch = max(self.config['samples_per_epoch'] // (self.config['batch_size'] * parallel.world_size()), 1)
        train_data_iter = None
        for i in range(iters_per_epoch):
            try:
                images, truths, name = next(train_data_iter)
            except (StopIteration, TypeError):
                train_data_iter = iter(self.training_data_loader)
                images, truths, name = next(train_data_iter)
            self.global_step = (self.current_epoch - 1) * iters_per_epoch + (i + 1) * self.config['batch_size']
            loss, output = self.train_step(images, truths)
            running_loss += loss.item()
            if i % self.config['print_batch_period'] == self.config['print_batch_period'] - 1:
                if parallel.rank() == 0:
                    print('Epoch[{}/{}] it {} loss: {:.3f} lr:{} {}'.format(
                        self.current_epoch, self.config['n_epochs'], i + 1,
                        running_loss / self.config['print_batch_period'] if i > 0 else running_loss,
                        self.optimizer.param_groups[0]['lr'], datetime.datetime.now().strftime("%D %H:%M:%S")))
                    if self.writer is not None:
                        self.writer.add_scalar('loss/training', running_loss / self.config['print_batch_period'], global_step=self.global_step)
                        self.writer.add_scalar('learning_rate', self.optimizer.param_groups[0]['lr'], global_step=self.global_step)
                running_loss = 0.0

    def eval(self, dataloader):
        """models/segmentation.py:179-201; Dice from exact integer counts computed on the device."""
        with torch.no_grad():
            self.model.eval()
            dice_per_class = torch.zeros(self.config["n_classes"] - 1, dtype=torch.float64)
            n_vol = 0
            pred = images = truths = None
            for j, (images, truths, name) in enumerate(dataloader):
                pred = self.model(images.to(self.device))
                d = metrics.metricEval('dice', pred, truths.to(self.device))       # [N][C-1]
                dice_per_class += torch.from_numpy(d.sum(0))
                n_vol += d.shape[0]                                                # average over VOLUMES (loaders may batch > 1)
            dice_per_class = (dice_per_class / max(n_vol, 1)).float()
            dice_avg = dice_per_class.mean()
            sample_for_vis = {'img': images, 'truth': truths, 'pred': pred}
        return dice_per_class, dice_avg, sample_for_vis

    def validate(self):
        if self.current_epoch % self.config['valid_epoch_period'] == 0:
            start_time = time.time()
            dice_per_class, dice_avg, samples = self.eval(self.validation_data_loader)
            if self.scheduler is not None:
                if self.config['lr_mode'] == 'plateau':
                    self.scheduler.step(dice_avg)
                else:
                    self.scheduler.step()
            is_best = False
            if dice_avg > self.best_score:
                is_best = True
                self.best_score = dice_avg
            if parallel.rank() != 0:
                return
            if self.writer is not None:
                self.writer.add_scalar('validation_{}/dice_avg'.format(self.config['data']), dice_avg, global_step=self.global_step)
            print("Validation: Dice Avg: {:.4f} ({:.3f} sec) {}".format(float(dice_avg), time.time() - start_time,
                                                                         datetime.datetime.now().strftime("%D %H:%M:%S")))
            if self.current_epoch % self.config['save_ckpts_epoch_period'] == 0:
                self.save_checkpoint({'epoch': self.current_epoch,
                                      'model_state_dict': self.model.state_dict(),
                                      'optimizer_state_dict': self.optimizer.state_dict(),
                                      'best_score': self.best_score},
                                     is_best, self.ckpoint_dir)

    def test(self, best=True, if_log=True):
        """models/segmentation.py:253-274: reload the best checkpoint and report Dice on the test loader."""
        self.setup_model()
        ckpoint_file = os.path.join(self.ckpoint_dir, 'model_best.pth.tar' if best else 'checkpoint.pth.tar')
        last_epoch, best_score = self.initialize_model(self.model, optimizer=None, ckpoint_path=ckpoint_file)
        loader = self.config.get('testing_data_loader') or self.validation_data_loader
        dice_per_class, dice_avg, samples = self.eval(loader)
        print('Testing Model: {} ({} epochs)  Dice_avg: {}'.format(ckpoint_file, last_epoch, float(dice_avg)))
        return dice_per_class, dice_avg
