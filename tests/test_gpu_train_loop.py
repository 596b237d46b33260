"""End-to-end: the reference's pre-processed file formats -> DaftExprtDataLoader -> collate -> Trainer (fused step,
gradient accumulation) -> checkpoint in the reference's format -> reload -> synthesis through generate.py."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import make_hparams


@pytest.mark.parametrize('ddp_flag', [False, True, 'collectives'])
def test_train_from_feature_files_then_checkpoint_and_synthesis(golden_dir, tmp_path, ddp_flag, monkeypatch):
    ''' ddp_flag=True is the default launch path of scripts/training.py on a 1-GPU node: `--multiprocessing_distributed`
        with world_size 1 -- process group of one rank, DistributedSampler, `module.`-prefixed checkpoints;
        'collectives': the same with DX_FORCE_DIST=1 -- the one-rank RCCL world issues every collective of the multi-rank loop
        (per-bucket all-reduce from the backward hooks, per-bucket Adam, validation reduction) '''
    if ddp_flag == 'collectives':
        monkeypatch.setenv('DX_FORCE_DIST', '1')
        ddp_flag = True
    from daft_exprt.generate import generate_mel_specs
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import train
    fx = np.load(os.path.join(golden_dir, 'data_loader.npz'))
    out = str(tmp_path)
    hp = make_hparams(training_files=os.path.join(golden_dir, 'train_list.txt'), validation_files=os.path.join(golden_dir, 'train_list.txt'),
                      output_directory=out, batch_size=2, accumulation_steps=2, nb_iterations=2, iters_per_checkpoint=2,
                      iters_check_for_model_improvement=1)
    hp.stats = {f'spk {i}': {'energy': {'mean': float(fx['stats_energy_mean'][i]), 'std': float(fx['stats_energy_std'][i])},
                             'pitch': {'mean': float(fx['stats_pitch_mean'][i]), 'std': float(fx['stats_pitch_std'][i])}} for i in range(11)}
    hp.rank, hp.world_size, hp.multiprocessing_distributed = 0, 1, ddp_flag
    hp.ngpus_per_node, hp.dist_url = 1, 'tcp://127.0.0.1:29517'
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        os.makedirs(os.path.join(out, 'logs'), exist_ok=True)
        train(0, hp, os.path.join(out, 'logs', 'train.log'))
    finally:
        os.chdir(cwd)
    assert not torch.distributed.is_initialized()
    allrecs = [json.loads(l) for l in open(os.path.join(out, 'logs', 'metrics.jsonl'))]
    recs = [r for r in allrecs if 'DaftExprt.training/loss' in r]
    vals = [r for r in allrecs if 'DaftExprt.validation/loss' in r]
    assert [r['iteration'] for r in recs] == [1, 2] and [r['iteration'] for r in vals] == [1, 2]
    assert all(math.isfinite(r['DaftExprt.validation/loss']) and 'DaftExprt.validation/mel_spec_l1_loss' in r for r in vals)
    best = torch.load(os.path.join(out, 'checkpoints', 'DaftExprt_best'), weights_only=False)   # train.py:446-451
    assert best['best_val_loss'] == pytest.approx(min(r['DaftExprt.validation/loss'] for r in vals))
    assert best['iteration'] in (1, 2) and len(best['state_dict']) == 193
    assert all(math.isfinite(r['DaftExprt.training/loss']) and r['DaftExprt.optimization/grad_norm'] > 0 for r in recs)
    ckpt_path = os.path.join(out, 'checkpoints', 'DaftExprt_2')
    ckpt = torch.load(ckpt_path, weights_only=False)
    assert ckpt['iteration'] == 2 and len(ckpt['state_dict']) == 193
    assert ckpt['best_val_loss'] == pytest.approx(best['best_val_loss'])
    assert all(k.startswith('module.') == ddp_flag for k in ckpt['state_dict'])
    # reload into a fresh model (what scripts/synthesize.py:38-44 does) and synthesise two sentences
    model = DaftExprt(hp)
    model.load_state_dict({k.replace('module.', ''): v for k, v in ckpt['state_dict'].items()})
    with torch.no_grad():   # untrained duration head: centre it so that the utterances have a sensible length
        model.prosody_predictor.projection.linear_layer.weight[0].mul_(0.05)
        model.prosody_predictor.projection.linear_layer.bias.copy_(torch.tensor([0.08, 0., 0.]))
    model = model.cuda(0)
    sentences = [[['HH', 'AH0', 'L', 'OW1'], ' ', ['W', 'ER1', 'L', 'D'], '.', '~'], [['T', 'EH1', 'S', 'T'], '~']]
    refs = [(fx['item0_frames_energy'], fx['item0_frames_pitch'], fx['item0_mel']), (fx['item1_frames_energy'], fx['item1_frames_pitch'], fx['item1_mel'])]
    preds = generate_mel_specs(model, sentences, ['a', 'b'], [0, 3], refs, os.path.join(out, 'synth'), hp, batch_size=2, get_time_perf=True)
    from daft_exprt.generate import LAST_TIME_PERF
    assert set(preds) == {'a_spk_0_ref_mem0', 'b_spk_3_ref_mem1'} and LAST_TIME_PERF['rtf'] > 0
    for name, (duration, duration_int, energy, pitch, mel_spec, alignment) in preds.items():
        assert mel_spec.shape[0] == 80 and mel_spec.shape[1] == int(duration_int.sum()) and np.isfinite(mel_spec).all()
        assert alignment.shape == (len(duration), mel_spec.shape[1])
        assert os.path.isfile(os.path.join(out, 'synth', f'{name}.npz'))
