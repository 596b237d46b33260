"""CPU tests of the host-side logic of the product package (no GPU, no oracle in the product path):
hyper-parameters, schedules, collate contracts vs the golden fixtures, state_dict ABI, checkpoint round trip."""
import json
import os

import numpy as np
import pytest
import torch

from daft_exprt.data_loader import DaftExprtDataCollate, SyntheticUtterances, synthetic_batch
from daft_exprt.hparams import HyperParams
from daft_exprt.loss import DaftExprtLoss
from daft_exprt.model import DaftExprt, param_table
from daft_exprt.train import update_learning_rate
from tests.util import make_hparams, INPUT_NAMES, COLLATE_NAMES, load_driver_fixture


def test_hparams_defaults_and_derived_fields(tmp_path):
    hp = make_hparams()
    assert hp.n_symbols == 76 and hp.symbols[0] == '_'
    assert hp.n_speakers == 12 and hp.speakers_id == list(range(11))          # hparams.py:195-203 of the reference
    assert hp.batch_size == 16 and hp.accumulation_steps == 3 and hp.betas == (0.9, 0.98) and hp.epsilon == 1e-9
    assert hp.prosody_encoder['attn_nb_heads'] == 8 and hp.frame_decoder['conv_channels'] == 1024
    assert hp.compute_dtype == 'bf16'
    with pytest.raises(AssertionError):
        HyperParams(verbose=False, training_files='a', validation_files='b', output_directory='c', language='english')  # speakers missing
    with pytest.raises(AssertionError):
        make_hparams(filter_length=1000)          # not a multiple of hop_length
    path = os.path.join(tmp_path, 'cfg', 'config.json')
    hp.save_hyper_params(path)
    cfg = json.load(open(path))
    hp2 = HyperParams(verbose=False, **cfg)       # JSON round trip, as launch_training does
    assert hp2.n_speakers == hp.n_speakers and hp2.prosody_encoder == hp.prosody_encoder and list(hp2.betas) == [0.9, 0.98]


def test_schedules_match_reference_kats(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'schedules.npz'))
    hp = make_hparams()
    crit = DaftExprtLoss(0, hp)
    for it, lr, adv in zip(fx['iterations'], fx['lr'], fx['adv']):
        assert update_learning_rate(hp, int(it)) == pytest.approx(float(lr), rel=1e-12)
        assert crit.update_adversarial_weight(int(it)) == pytest.approx(float(adv), rel=1e-12)


def test_collate_contract_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'forward_eval.npz'))
    items = []
    for i in range(4):
        g = lambda nm: torch.from_numpy(fx[f'item{i}_{nm}'])
        items.append([g('symbols'), g('dur_float'), g('dur_int'), g('sym_energy'), g('sym_pitch'), g('frames_energy'),
                      g('frames_pitch'), g('mel'), int(fx[f'item{i}_speaker']), f'dir{i}', f'file{i}'])
    batch = DaftExprtDataCollate(make_hparams())(items)
    for name, got in zip(INPUT_NAMES, batch[:11]):
        want = fx[f'in_{name}']
        assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want), name
    assert list(batch[11]) == list(fx['collate_dirs']) and list(batch[12]) == list(fx['collate_files'])


@pytest.mark.parametrize('transform', ['add', 'multiply'])
def test_inference_collate_replays_the_reference_bit_exact(golden_dir, transform, tmp_path):
    ''' raw driver inputs (nested sentences, per-symbol factor lists, reference .npz files) -> the product's
        `generate.collate_tensors` must give exactly what the reference's gave (`generate.py:140-239`), paths or triples '''
    import json
    from daft_exprt.generate import collate_tensors
    hp = make_hparams()
    for ref_dir in (None, str(tmp_path)):
        sentences, dur_f, en_f, pi_f, refs, spk, names, fx = load_driver_fixture(golden_dir, transform, ref_dir)
        col = collate_tensors(sentences, dur_f, en_f, pi_f, transform, refs, spk, list(names), hp)
        for nm, got in zip(COLLATE_NAMES, col[:-1]):
            want = fx[f'{transform}_col_{nm}']
            assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want), nm
        assert list(col[-1]) == json.loads(str(fx[f'{transform}_col_file_names_json']))
    with pytest.raises(AssertionError):
        collate_tensors(sentences, [[1.]] + dur_f[1:], en_f, pi_f, transform, refs, spk, list(names), hp)   # wrong factor count


def test_synthetic_batches_follow_the_collate_invariants():
    hp = make_hparams()
    b = synthetic_batch(hp, 6, seed=3, t_max=300)
    symbols, dur_f, dur_i, s_en, s_pi, in_len, f_en, f_pi, mel, out_len, spk = b[:11]
    assert torch.all(in_len[:-1] >= in_len[1:])                                  # sorted by phoneme count
    assert torch.equal(dur_i.sum(1), out_len) and int(out_len.max()) == mel.shape[2] == 300
    assert int(symbols.max()) < hp.n_symbols and int(spk.max()) < hp.n_speakers - 1
    for r in range(6):
        assert not symbols[r, in_len[r]:].any() and not mel[r, :, out_len[r]:].any()   # zero padding
    ds = SyntheticUtterances(hp, 4, seed=3)
    assert torch.equal(ds[2][0], ds[2][0])                                       # deterministic per index


def test_state_dict_abi_and_module_prefix():
    hp = make_hparams()
    m = DaftExprt(hp)
    sd = m.state_dict()
    assert len(sd) == 193 and sum(v.numel() for v in sd.values()) == 14727153
    assert sd['prosody_encoder.post_multipliers'].shape == (2, 9)
    assert sd['frame_decoder.blocks.3.attention.multi_head_attention.in_proj_weight'].shape == (384, 128)
    assert sd['prosody_predictor.projection.linear_layer.weight'].shape == (3, 256)
    assert sd['gaussian_upsampling.projection.0.linear_layer.weight'].shape == (1, 128)
    assert sd['prosody_encoder.gammas_predictor.linear_layer.weight'].shape == (1280, 128)
    # parameters are views of one flat buffer, in registration order
    flat = m.flat_parameters()
    off = 0
    for (name, shape, _), (n2, p) in zip(param_table(hp), m.named_parameters()):
        assert name == n2 and p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    # load_state_dict keeps the views
    m2 = DaftExprt(hp)
    m2.load_state_dict({k: v.clone() for k, v in sd.items()})
    assert torch.equal(m2.flat_parameters(), flat) and m2.prosody_encoder.post_multipliers.data_ptr() == m2.flat_parameters().data_ptr()


def test_checkpoint_round_trip(tmp_path):
    from daft_exprt.optim import FusedAdam
    from daft_exprt.train import save_checkpoint, load_checkpoint
    import torch.distributed as dist
    hp = make_hparams()
    m = DaftExprt(hp)
    opt = FusedAdam(m, lr=3e-4)
    plain = os.path.join(tmp_path, 'checkpoints', 'plain')
    save_checkpoint(m, opt, hp, 3e-4, 1, best_val_loss=2., filepath=plain)
    assert not any(k.startswith('module.') for k in torch.load(plain, weights_only=False)['state_dict'])
    # a live process group (any world size, like the reference's DDP wrap) -> 'module.' prefixed keys
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29533', world_size=1, rank=0)
    try:
        _round_trip(tmp_path, hp, m, opt, save_checkpoint, load_checkpoint, FusedAdam)
    finally:
        dist.destroy_process_group()


def _round_trip(tmp_path, hp, m, opt, save_checkpoint, load_checkpoint, FusedAdam):
    opt.step_count = 7
    opt.exp_avg.uniform_(-1, 1)
    opt.exp_avg_sq.uniform_(0, 1)
    path = os.path.join(tmp_path, 'checkpoints', 'DaftExprt_7')
    save_checkpoint(m, opt, hp, 3e-4, 7, best_val_loss=1.5, filepath=path)
    ckpt = torch.load(path, weights_only=False)
    assert set(ckpt) == {'iteration', 'learning_rate', 'best_val_loss', 'state_dict', 'optimizer', 'config_params'}   # train.py:73-78
    assert all(k.startswith('module.') for k in ckpt['state_dict'])
    assert set(ckpt['optimizer']['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}                                     # torch Adam layout
    m2 = DaftExprt(hp)
    opt2 = FusedAdam(m2)
    _, _, it, lr, best = load_checkpoint(path, 0, m2, opt2, hp)
    assert (it, lr, best) == (7, 3e-4, 1.5) and opt2.step_count == 7
    assert torch.equal(m2.flat_parameters(), m.flat_parameters()) and torch.equal(opt2.exp_avg, opt.exp_avg)


def test_product_fails_loudly_without_gpu():
    ''' no CPU fallback: CPU tensors must raise, never silently run elsewhere '''
    hp = make_hparams()
    m = DaftExprt(hp)
    batch = synthetic_batch(hp, 2, seed=1, t_max=40, l_range=(5, 9))
    inputs = tuple(t for t in batch[:11])
    with pytest.raises(RuntimeError):
        m(inputs)


def test_feature_reader_matches_reference(golden_dir):
    ''' the on-disk feature reader vs the reference's DaftExprtDataLoader on the same files '''
    from daft_exprt.data_loader import DaftExprtDataLoader
    fx = np.load(os.path.join(golden_dir, 'data_loader.npz'))
    hp = make_hparams()
    hp.stats = {f'spk {i}': {'energy': {'mean': float(fx['stats_energy_mean'][i]), 'std': float(fx['stats_energy_std'][i])},
                             'pitch': {'mean': float(fx['stats_pitch_mean'][i]), 'std': float(fx['stats_pitch_std'][i])}} for i in range(11)}
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        ds = DaftExprtDataLoader(os.path.join(golden_dir, 'train_list.txt'), hp, shuffle=True)
        assert len(ds) == int(fx['n_items'])
        for i in range(len(ds)):
            item = ds[i]
            assert item[10] == str(fx[f'item{i}_file'])            # same seeded shuffle order
            assert item[8] == int(fx[f'item{i}_speaker'])
            for j, nm in enumerate(['symbols', 'dur_float', 'dur_int', 'sym_energy', 'sym_pitch', 'frames_energy', 'frames_pitch', 'mel']):
                want = fx[f'item{i}_{nm}']
                got = item[j].numpy()
                assert got.dtype == want.dtype and np.array_equal(got, want), (i, nm)
        batch = DaftExprtDataCollate(hp)([ds[i] for i in range(len(ds))])
        assert batch[8].shape == (5, 80, 30) and batch[5].tolist() == sorted(batch[5].tolist(), reverse=True)
    finally:
        os.chdir(cwd)


def test_unsupported_widths_and_bad_ids_are_rejected_up_front():
    hp = make_hparams()
    hp.phoneme_encoder['hidden_embed_dim'] = 256
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)
    hp = make_hparams()
    hp.prosody_encoder['attn_nb_heads'] = 3      # (8, 4, 2, 1 heads have kernels)
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)
    hp = make_hparams()
    m = DaftExprt(hp)
    batch = list(synthetic_batch(hp, 3, seed=3, t_max=40, l_range=(4, 9)))
    bad = list(batch)
    bad[0] = batch[0].clone()
    bad[0][0, 0] = hp.n_symbols                     # nn.Embedding would raise
    with pytest.raises(IndexError):
        m.parse_batch('cuda:0', tuple(bad))
    bad = list(batch)
    bad[10] = torch.full_like(batch[10], hp.n_speakers - 1)    # CrossEntropyLoss: target out of bounds
    with pytest.raises(IndexError):
        m.parse_batch('cuda:0', tuple(bad))
    m.check_ids(batch[0], torch.full_like(batch[10], hp.n_speakers - 1), training=False)   # a valid embedding row at inference


def test_dropout_seeds_differ_between_ranks():
    hp = make_hparams()
    a, b = DaftExprt(hp), DaftExprt(hp)
    b.set_rank(1)
    a._step_id = b._step_id = 5
    sa = [a._seed() for _ in range(4)]
    sb = [b._seed() for _ in range(4)]
    assert len(set(sa + sb)) == 8 and all(0 <= v < 2 ** 63 for v in sa + sb)


def test_grouped_micro_batches_host_side():
    ''' `data_loader.group_host_batches` / `group_micro_batches`: the micro-batches of an optimizer step concatenated and padded to the
        group's maxima, every utterance carrying the padded length of its OWN micro-batch (n_max) and the skip length
        min(len, n_max - 2) the kernels need; contents of every micro-batch preserved, padding zero '''
    from daft_exprt.data_loader import group_host_batches, group_micro_batches, synthetic_batch
    hp = make_hparams(batch_size=3, accumulation_steps=3)
    bs = [synthetic_batch(hp, 3, seed=10 + k, t_max=t, force_first_full=True, l_range=(5 + 4 * k, 12 + 4 * k)) for k, t in enumerate((90, 40, 61))]
    merged, (nmax_in, nmax_out), sizes = group_host_batches(bs)
    assert sizes == [3, 3, 3] and len(merged) == 13 and len(merged[11]) == 9 and len(merged[12]) == 9
    Lg, Tg = max(b[0].shape[1] for b in bs), 90
    assert merged[0].shape == (9, Lg) and merged[8].shape == (9, 80, Tg) and merged[6].shape == (9, Tg)
    assert nmax_out.tolist() == [90] * 3 + [40] * 3 + [61] * 3
    assert nmax_in.tolist() == sum(([b[0].shape[1]] * 3 for b in bs), [])
    row = 0
    for b in bs:
        L, T = b[0].shape[1], b[8].shape[2]
        assert torch.equal(merged[0][row:row + 3, :L], b[0]) and not merged[0][row:row + 3, L:].any()
        assert torch.equal(merged[8][row:row + 3, :, :T], b[8]) and not merged[8][row:row + 3, :, T:].any()
        assert torch.equal(merged[9][row:row + 3], b[9]) and torch.equal(merged[5][row:row + 3], b[5])
        row += 3
    g = group_micro_batches([(b[:11], None) for b in bs])
    skip_in, n_in, skip_out, n_out = g.bounds
    assert torch.equal(n_out, nmax_out) and torch.equal(n_in, nmax_in) and g.accum == 3
    assert torch.equal(skip_out, torch.clamp(torch.minimum(merged[9], nmax_out - 2), min=0))
    assert int((merged[9] - skip_out).max()) == 2        # the utterance that fills its micro-batch: rows len - 2 .. are its last live ones
    assert all(t is u for t, u in zip(g.targets, (g.inputs[1], g.inputs[3], g.inputs[4], g.inputs[8], g.inputs[10])))
