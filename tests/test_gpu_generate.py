"""The inference driver (`daft_exprt/generate.py`) against what the REFERENCE's driver produced from the same raw inputs
(tests/golden/inference_collate.npz, written by tools/gen_goldens.py section C2 from `generate.generate_mel_specs` of the
imported reference, closed-form fill weights): prediction keys (`<name>_spk_<id>_ref_<reference>`) in the same order,
list-of-6 values, `.npz` files holding only `mel_spec`, the caller's name list untouched, RTF accounting.
fp32 operand mode: integer durations bit-exact, floats within 2e-4 of each tensor's max."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O
from oracle.fill import fill_params
from tests.util import make_hparams, load_driver_fixture


@pytest.mark.parametrize('transform', ['add', 'multiply'])
def test_driver_matches_reference_driver(golden_dir, tmp_path, transform):
    from daft_exprt import generate as G
    from daft_exprt.model import DaftExprt
    st = np.load(os.path.join(golden_dir, 'inference.npz'))
    hp = make_hparams(compute_dtype='fp32')
    hp.stats = {f'spk {i}': {'pitch': {'mean': float(st['stats_pitch_mean'][i]), 'std': float(st['stats_pitch_std'][i])}}
                for i in range(11)}
    model = DaftExprt(hp)
    model.load_state_dict(fill_params(O.param_shapes(hp)))
    model = model.cuda(0)
    ref_dir, out_dir = str(tmp_path / 'refs'), str(tmp_path / 'out')
    os.makedirs(ref_dir)
    sentences, dur_f, en_f, pi_f, refs, spk, names, fx = load_driver_fixture(golden_dir, transform, ref_dir)
    mine = list(names)
    preds = G.generate_mel_specs(model, sentences, mine, spk, refs, out_dir, hp, dur_factors=dur_f, energy_factors=en_f,
                                 pitch_factors=[transform.upper(), pi_f], batch_size=2, n_jobs=1, use_griffin_lim=False,
                                 get_time_perf=True)
    assert isinstance(preds, dict)                                                   # get_time_perf does not change the type
    assert list(preds.keys()) == json.loads(str(fx[f'{transform}_drv_keys_json']))
    assert mine == json.loads(str(fx[f'{transform}_drv_names_after_json']))
    files = sorted(os.listdir(out_dir))
    assert files == json.loads(str(fx[f'{transform}_drv_files_json']))
    assert sorted(np.load(os.path.join(out_dir, files[0])).files) == json.loads(str(fx[f'{transform}_drv_npz_keys_json']))
    audio = 0.
    for k, (key, vals) in enumerate(preds.items()):
        assert isinstance(vals, list) and len(vals) == 6
        for nm, got in zip(['duration', 'duration_int', 'energy', 'pitch', 'mel_spec', 'alignment'], vals):
            want = fx[f'{transform}_drv{k}_{nm}']
            assert got.shape == want.shape and got.dtype == want.dtype, (key, nm, got.shape, want.shape, got.dtype, want.dtype)
            if nm == 'duration_int':
                assert np.array_equal(got, want), (key, got, want)
            else:
                assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-6, (key, nm, np.abs(got - want).max())
        assert np.array_equal(np.load(os.path.join(out_dir, f'{key}.npz'))['mel_spec'], vals[4])
        audio += ((vals[4].shape[1] - 1) * hp.hop_length + hp.filter_length - 2 * int(hp.filter_length / 2)) / hp.sampling_rate
    perf = G.LAST_TIME_PERF
    assert perf['sentences'] == 5 and perf['audio_seconds'] == pytest.approx(audio) and perf['wall_seconds'] > 0
    assert perf['rtf'] == pytest.approx(audio / perf['wall_seconds'])
    with pytest.raises(AssertionError):
        G.generate_mel_specs(model, sentences, list(names), spk, refs, out_dir, hp, pitch_factors=['scale', pi_f])
    with pytest.raises(AssertionError):
        G.generate_mel_specs(model, sentences, list(names)[:-1], spk, refs, out_dir, hp)
