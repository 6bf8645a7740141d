"""Train a 3D U-Net: the build's counterpart of the reference's train_seg.py (same flags, same config dict,
train_seg.py:10-93) with its three shipped bugs fixed: the em-dash in '--num-epochs', the
positional './data' / './logs' defaults, and the undefined args.leaf.  Device selection is explicit and the
data source is the synthetic volume generator (the MindBoggle NIfTI pipeline needs SimpleITK)."""
import argparse
import os

from deepatlas_amd.models.segmentation import SegmentationExperiment


def build_config(args):
    n_classes = 32
    config = dict(
        debug_mode=args.debug,
        resume_dir='',
        random_seed=230,
        data='synthetic',
        n_epochs=args.num_epochs,
        samples_per_epoch=args.num_samples * 2,   # due to flipping data augmentation
        batch_size=1,
        valid_batch_size=1,
        print_batch_period=50,
        valid_epoch_period=1,
        save_ckpts_epoch_period=1,
        model='UNet_light',
        model_settings={'in_channel': 1, 'n_classes': n_classes, 'bias': True, 'BN': True},
        n_classes=n_classes,
        class_name={k: str(k) for k in range(1, n_classes)},
        crop_size=[0, 10, 7, 14, 8, 7],
        loss='dice',
        loss_settings={'n_class': n_classes, 'weight_type': 'Uniform', 'no_bg': False, 'softmax': True, 'eps': 1e-6},
        learning_rate=1e-3,
        lr_mode='multiStep',
        milestones=[0.5, 1],
        gamma=0.2,
    )
    config.update(args.__dict__)
    config['learning_rate'] = args.lr
    config['synthetic_shape'] = tuple(args.shape)
    config['data_dir'] = os.path.join(args.data_root, "synthetic")
    config['valid_data_dir'] = config['data_dir']
    config['log_dir'] = './{}/{}'.format(args.log_root, config['data'])
    config['device'] = 'cuda:{}'.format(args.device) if args.device.isdigit() else args.device
    if not config.get('matrix_precision'):                # (--matrix-precision; not a key of the reference's config: absent = the package default, 'fp32_split')
        config.pop('matrix_precision', None)
    return config


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--device', '-g', default='0', type=str, help='index of used GPU')
    parser.add_argument('--debug', '-d', action='store_true', help='if debug mode')
    parser.add_argument('--preload', '-load', action='store_true', help='if preload data into memory to speed up IO')
    parser.add_argument('--num-samples', '-ns', default=21, type=int, help='number of samples for training')
    parser.add_argument('--num-epochs', '-ne', default=100, type=int, help='number of epochs for training')
    parser.add_argument('--lr', default=1e-3, type=float, help='learning rate')
    parser.add_argument('--test_only', '-t', action='store_true', help='only test model')
    parser.add_argument('--data-root', '-root', default='./data', type=str, help='root of the data folder')
    parser.add_argument('--log-root', '-log', default='./logs', type=str, help='root of the log folders')
    parser.add_argument('--shape', nargs=3, type=int, default=[64, 64, 64], help='synthetic volume size D H W (multiples of 8)')
    parser.add_argument('--matrix-precision', default=None, choices=['fp32', 'fp32_split', 'bf16'],
                        help="arithmetic of the 3x3x3 convolutions: 'fp32_split' (default; what bench.py measures) every fp32 operand scaled per staged "
                             "tile and split into two fp16 terms, three products per multiply on the fp16 matrix pipe: 22-bit products (per-product bound "
                             "7e-7, NARROWER than an fp32 multiply; sums of >= ~100 terms are as close to double as fp32's; elements > 2^15..2^18 below "
                             "their tile's maximum keep only an absolute 2^-40 of it -- csrc/split_f16.h); 'fp32' the fp32 matrix instructions = the "
                             "reference's arithmetic (~1.7x slower steps); 'bf16' operands rounded to bf16")
    args = parser.parse_args(argv)
    exp = SegmentationExperiment(build_config(args))
    if not args.test_only:
        exp.train()
    exp.test()


if __name__ == '__main__':
    main()
