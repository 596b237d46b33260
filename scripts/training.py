#!/usr/bin/env python
"""CLI of the training job, same flags as the reference `scripts/training.py:131-203` for the `train` sub-command:
    python scripts/training.py -en EXP -dd DATA_DIR -spks SPK [SPK ...] -lg english train [-chk CKPT] [-nmpd] [-ws N] [-r R] [-m URL]
It builds `HyperParams`, writes `<experiment>/config.json` and runs `daft_exprt/train.py` in a sub-process, like the
reference (`training.py:101-116`).  `pre_process` / `fine_tune` are dataset tooling outside the accelerated path."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')
sys.path.insert(0, PKG)

from daft_exprt.hparams import HyperParams  # noqa: E402


def train(args, hparams, config_file, log_file):
    cmd = [sys.executable, os.path.join(PKG, 'daft_exprt', 'train.py'), '--data_set_dir', args.data_set_dir, '--config_file', config_file,
           '--benchmark_dir', os.path.join(ROOT, 'scripts', 'benchmarks'), '--log_file', log_file, '--world_size', str(args.world_size),
           '--rank', str(args.rank), '--master', args.master]
    if not args.no_multiprocessing_distributed:
        cmd.append('--multiprocessing_distributed')
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''))
    return subprocess.call(cmd, env=env)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='script to train Daft-Exprt on MI355X')
    parser.add_argument('-en', '--experiment_name', type=str, required=True)
    parser.add_argument('-dd', '--data_set_dir', type=str, required=True)
    parser.add_argument('-spks', '--speakers', nargs='*', default=[])
    parser.add_argument('-lg', '--language', type=str, default='english')
    sub = parser.add_subparsers(dest='command')
    p_train = sub.add_parser('train')
    p_train.add_argument('-chk', '--checkpoint', type=str, default='')
    p_train.add_argument('-nmpd', '--no_multiprocessing_distributed', action='store_true')
    p_train.add_argument('-ws', '--world_size', type=int, default=1)
    p_train.add_argument('-r', '--rank', type=int, default=0)
    p_train.add_argument('-m', '--master', type=str, default='tcp://localhost:54321')
    for name in ('pre_process', 'fine_tune'):
        sub.add_parser(name)
    args = parser.parse_args()
    if args.command != 'train':
        sys.exit(f'"{args.command}" is dataset tooling of the reference (MFA / librosa / REAPER); only "train" is accelerated here')
    out_dir = os.path.join(ROOT, 'trainings', args.experiment_name)
    features_dir = os.path.join(ROOT, 'datasets', args.language, '22050Hz')
    hparams = HyperParams(training_files=os.path.join(features_dir, f'train_{args.language}.txt'),
                          validation_files=os.path.join(features_dir, f'validation_{args.language}.txt'), output_directory=out_dir,
                          language=args.language, speakers=args.speakers, checkpoint=args.checkpoint)
    config_file = os.path.join(out_dir, 'config.json')
    hparams.save_hyper_params(config_file)
    os.makedirs(os.path.join(out_dir, 'logs'), exist_ok=True)
    sys.exit(train(args, hparams, config_file, os.path.join(out_dir, 'logs', 'train.log')))
