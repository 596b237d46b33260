#!/usr/bin/env python
"""Checkpoint interop against the reference's OWN code, run in the build container (where /root/reference exists; the test that
calls this skips elsewhere -- nothing here travels to the GPU box):

  1. the reference's DaftExprt (default architecture, 11 speakers) takes two `torch.optim.Adam` steps (the reference trainer's
     settings, train.py:299-301) on synthetic gradients and writes a checkpoint with ITS `save_checkpoint` (train.py:56-78);
  2. this package's `load_checkpoint` reads that file into `DaftExprt` + `FusedAdam`: every one of the 193 tensors, both Adam
     moments of each, the step count, iteration / learning rate / best validation loss must come back bit for bit;
  3. the way back: this package's `save_checkpoint` (with and without the `module.` prefix of a data-parallel run) is read by the
     reference's `load_checkpoint` (train.py:81-136) into the reference model and a fresh `torch.optim.Adam`.
Prints `REF_CHECKPOINT_OK <n tensors>`."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    import gen_goldens as G
    G.install_shims()                                     # stubs for absent third-party imports, .cuda() -> identity (CPU container)
    import daft_exprt.hparams as ref_hparams
    import daft_exprt.model as ref_model
    import daft_exprt.train as ref_train
    ref = {k: v for k, v in sys.modules.items() if k == 'daft_exprt' or k.startswith('daft_exprt.')}
    hp_ref = G.make_hparams(ref_hparams)
    torch.manual_seed(3)
    m_ref = ref_model.DaftExprt(hp_ref)
    opt_ref = torch.optim.Adam(m_ref.parameters(), lr=hp_ref.initial_learning_rate if hasattr(hp_ref, 'initial_learning_rate') else 1e-4,
                               betas=hp_ref.betas, eps=hp_ref.epsilon, weight_decay=hp_ref.weight_decay, amsgrad=False)
    g = torch.Generator().manual_seed(5)
    for _ in range(2):
        for p in m_ref.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        opt_ref.step()
    tmp = tempfile.mkdtemp(prefix='dx_ref_ckpt_')
    path = os.path.join(tmp, 'checkpoints', 'DaftExprt_2')
    _orig_map = torch.load

    def cpu_load(f, map_location=None, **kw):             # the reference maps to f'cuda:{gpu}' (train.py:96); this container has no GPU
        kw.setdefault('weights_only', False)
        return _orig_map(f, map_location='cpu', **kw)
    ref_train.save_checkpoint(m_ref, opt_ref, hp_ref, 3.3e-4, 2, best_val_loss=1.25, filepath=path)
    assert os.path.isfile(path)
    ref_state = {k: v.clone() for k, v in m_ref.state_dict().items()}
    ref_opt = opt_ref.state_dict()

    # ---- this package (its `daft_exprt` shadows the reference's: swap the module tables)
    for k in list(ref):
        del sys.modules[k]
    sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
    from daft_exprt.hparams import HyperParams
    from daft_exprt.model import DaftExprt
    from daft_exprt.optim import FusedAdam
    from daft_exprt import train as T
    mine_mods = {k: v for k, v in sys.modules.items() if k == 'daft_exprt' or k.startswith('daft_exprt.')}
    hp = HyperParams(verbose=False, training_files='none', validation_files='none', output_directory=tmp, language='english',
                     speakers=list(G.SPEAKERS))
    torch.manual_seed(99)                                  # different initial weights: everything must come from the file
    model = DaftExprt(hp)
    opt = FusedAdam(model, betas=hp.betas, eps=hp.epsilon, weight_decay=hp.weight_decay)
    model, opt, iteration, lr, best = T.load_checkpoint(path, 0, model, opt, hp)
    assert (iteration, lr, best) == (2, 3.3e-4, 1.25), (iteration, lr, best)
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref_state.keys()) and len(sd) == 193
    for k in sd:
        assert sd[k].dtype == ref_state[k].dtype and torch.equal(sd[k].cpu(), ref_state[k]), k
    assert opt.step_count == 2
    mine_opt = opt.state_dict()
    assert len(mine_opt['state']) == len(ref_opt['state']) == 193
    for idx, st in ref_opt['state'].items():
        for key in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(mine_opt['state'][idx][key].cpu(), st[key]), (idx, key)
        assert float(mine_opt['state'][idx]['step']) == float(st['step']) == 2.
    for key in ('lr', 'betas', 'eps', 'weight_decay', 'amsgrad'):
        assert tuple(mine_opt['param_groups'][0][key]) == tuple(ref_opt['param_groups'][0][key]) if key == 'betas' else \
            mine_opt['param_groups'][0][key] == ref_opt['param_groups'][0][key], key

    # ---- the way back: files written by this package, read by the reference's load_checkpoint
    back = os.path.join(tmp, 'checkpoints', 'DaftExprt_back')
    T.save_checkpoint(model, opt, hp, 4.4e-4, 7, best_val_loss=0.5, filepath=back)
    for k in list(mine_mods):
        del sys.modules[k]
    sys.path.remove(os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
    sys.modules.update(ref)
    torch.load = cpu_load
    try:
        torch.manual_seed(123)
        m2 = ref_model.DaftExprt(hp_ref)
        o2 = torch.optim.Adam(m2.parameters(), lr=1e-4, betas=hp_ref.betas, eps=hp_ref.epsilon, weight_decay=hp_ref.weight_decay)
        m2, o2, it2, lr2, best2 = ref_train.load_checkpoint(back, 0, m2, o2, hp_ref)
    finally:
        torch.load = _orig_map
    assert (it2, lr2, best2) == (7, 4.4e-4, 0.5)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, ref_state[k]), k
    st2 = o2.state_dict()['state']
    for idx, st in ref_opt['state'].items():
        assert torch.equal(st2[idx]['exp_avg'], st['exp_avg']) and torch.equal(st2[idx]['exp_avg_sq'], st['exp_avg_sq']), idx
    print('REF_CHECKPOINT_OK', len(sd))


if __name__ == '__main__':
    main()
