"""Batch construction for the trainer.

`DaftExprtDataCollate` keeps the collate contract of the reference (`src/daft_exprt/data_loader.py:146-211`):
sort by phoneme count (descending), right-zero-pad to the batch maxima, `output_lengths` follows the same
permutation (NOT sorted by T), 13-tuple order consumed by `DaftExprt.parse_batch`.
`SyntheticUtterances` generates the seeded synthetic utterances of SURVEY 8(d) (there are no corpora in the
build / bench environment).  `DaftExprtDataLoader` reads the reference's pre-processed feature files.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler


def _read_floats(path):
    with open(path, 'r', encoding='utf-8') as f:
        return np.array([float(line) for line in f.read().split()], dtype=np.float64)


class DaftExprtDataLoader(Dataset):
    ''' reader of the reference's pre-processed feature files (`data_loader.py:11-137`).  One list line
        `features_dir|feature_file|speaker_id` per utterance; per utterance: `<file>.npy` mel-spec (n_mel, T),
        `<file>.markers` TSV rows `begin end int_dur symbol word word_idx`, and one-float-per-line text files
        `.symbols_nrg .frames_nrg .symbols_f0 .frames_f0`.  Symbol-level energy / pitch are standardised with the
        speaker statistics of `hparams.stats`, zeros (unvoiced / silent) are preserved; frame-level values stay raw.
        The list is shuffled once with `random.seed(hparams.seed)` like the reference. '''
    def __init__(self, data_file, hparams, shuffle=True):
        assert os.path.isfile(data_file), f'no such list file "{data_file}"'
        with open(data_file, 'r', encoding='utf-8') as f:
            self.data = [line.strip().split('|') for line in f if line.strip()]
        self.hparams = hparams
        if shuffle:
            random.seed(hparams.seed)
            random.shuffle(self.data)

    def _standardise(self, values, speaker_id, feature):
        stats = self.hparams.stats[f'spk {speaker_id}'][feature]
        zeros = values == 0.
        values = (values - stats['mean']) / stats['std']
        values[zeros] = 0.
        return values

    def __getitem__(self, index):
        features_dir, feature_file, speaker_id = self.data[index][0], self.data[index][1], int(self.data[index][2])
        base = os.path.join(features_dir, feature_file)
        mel_spec = torch.from_numpy(np.load(base + '.npy'))
        assert mel_spec.size(0) == self.hparams.n_mel_channels
        symbols, durations_float, durations_int = [], [], []
        with open(base + '.markers', 'r', encoding='utf-8') as f:
            for line in f:
                begin, end, int_dur, symbol, _, _ = line.strip().split('\t')
                symbols.append(self.hparams.symbols.index(symbol))
                durations_float.append(float(end) - float(begin))
                durations_int.append(int(int_dur))
        symbols = torch.IntTensor(symbols)
        durations_float, durations_int = torch.FloatTensor(durations_float), torch.IntTensor(durations_int)
        symbols_energy = torch.FloatTensor(self._standardise(_read_floats(base + '.symbols_nrg'), speaker_id, 'energy'))
        symbols_pitch = torch.FloatTensor(self._standardise(_read_floats(base + '.symbols_f0'), speaker_id, 'pitch'))
        frames_energy = torch.FloatTensor(_read_floats(base + '.frames_nrg'))
        frames_pitch = torch.FloatTensor(_read_floats(base + '.frames_f0'))
        T = mel_spec.size(1)
        assert len(symbols_energy) == len(symbols) == len(symbols_pitch)
        assert len(frames_energy) == T == len(frames_pitch) and int(durations_int.sum()) == T
        return symbols, durations_float, durations_int, symbols_energy, symbols_pitch, frames_energy, frames_pitch, mel_spec, \
            speaker_id, features_dir, feature_file

    def __len__(self):
        return len(self.data)


def prepare_data_loaders(hparams, num_workers=1, drop_last=True, distributed=None):
    ''' `data_loader.py:214-243`: train / validation loaders; `DistributedSampler(shuffle=False)` when distributed
        (default: whenever a process group is live -- the reference keys on `hparams.multiprocessing_distributed`, which
        it only sets together with an initialised group) '''
    if distributed is None:
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    train_set = DaftExprtDataLoader(hparams.training_files, hparams)
    val_set = DaftExprtDataLoader(hparams.validation_files, hparams)
    collate_fn = DaftExprtDataCollate(hparams)
    sampler = DistributedSampler(train_set, shuffle=False) if distributed else None
    train_loader = DataLoader(train_set, num_workers=num_workers, shuffle=(sampler is None), sampler=sampler,
                              batch_size=hparams.batch_size, pin_memory=True, drop_last=drop_last, collate_fn=collate_fn)
    val_loader = DataLoader(val_set, num_workers=num_workers, shuffle=False, batch_size=hparams.batch_size, pin_memory=True,
                            drop_last=False, collate_fn=collate_fn)
    return train_loader, sampler, val_loader, len(train_set)


class DaftExprtDataCollate():
    def __init__(self, hparams):
        self.hparams = hparams

    def __call__(self, batch):
        ''' batch: list of [symbols, durations_float, durations_int, symbols_energy, symbols_pitch, frames_energy,
            frames_pitch, mel_spec (n_mel, T), speaker_id, features_dir, feature_file] '''
        n = len(batch)
        input_lengths, order = torch.sort(torch.LongTensor([len(x[0]) for x in batch]), dim=0, descending=True)
        L = int(input_lengths[0])
        T = max(x[7].size(1) for x in batch)
        symbols = torch.zeros(n, L, dtype=torch.long)
        durations_float = torch.zeros(n, L)
        durations_int = torch.zeros(n, L, dtype=torch.long)
        symbols_energy, symbols_pitch = torch.zeros(n, L), torch.zeros(n, L)
        frames_energy, frames_pitch = torch.zeros(n, T), torch.zeros(n, T)
        mel_specs = torch.zeros(n, self.hparams.n_mel_channels, T)
        output_lengths, speaker_ids = torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)
        feature_dirs, feature_files = [], []
        for row, src in enumerate(order.tolist()):
            item = batch[src]
            l, t = len(item[0]), item[7].size(1)
            symbols[row, :l] = item[0]
            durations_float[row, :l] = item[1]
            durations_int[row, :l] = item[2]
            symbols_energy[row, :l] = item[3]
            symbols_pitch[row, :l] = item[4]
            frames_energy[row, :t] = item[5]
            frames_pitch[row, :t] = item[6]
            mel_specs[row, :, :t] = item[7]
            output_lengths[row] = t
            speaker_ids[row] = item[8]
            feature_dirs.append(item[9])
            feature_files.append(item[10])
        return symbols, durations_float, durations_int, symbols_energy, symbols_pitch, input_lengths, \
            frames_energy, frames_pitch, mel_specs, output_lengths, speaker_ids, feature_dirs, feature_files


class SyntheticUtterances(Dataset):
    ''' seeded synthetic utterances with the statistics of SURVEY 8(d): L ~ U{40..160}, integer durations
        U{0..12} trimmed so that T <= t_max, mel ~ clip(N(-5, 2), ln 1e-5, 2), 30 % unvoiced frames. '''
    def __init__(self, hparams, n_items, seed=1234, t_min=1, t_max=1000, force_first_full=True, n_speakers=None,
                 l_range=(40, 160)):
        self.hp, self.n, self.seed = hparams, n_items, seed
        self.t_min, self.t_max, self.force_first_full = t_min, t_max, force_first_full
        self.n_speakers = n_speakers if n_speakers is not None else max(1, hparams.n_speakers - 1)
        self.l_range = l_range

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        rng = np.random.RandomState((self.seed * 1000003 + idx) % (2 ** 31 - 1))
        L = int(rng.randint(self.l_range[0], self.l_range[1] + 1))
        d = rng.randint(0, 13, size=L).astype(np.int64)
        if self.force_first_full and idx == 0:
            target = self.t_max
        else:
            target = int(np.clip(d.sum(), max(self.t_min, L // 2), self.t_max))
        while d.sum() > target:          # decrement the largest first
            d[int(np.argmax(d))] -= 1
        while d.sum() < target:
            d[int(rng.randint(0, L))] += 1
        T = int(d.sum())
        dur_f = (d * float(self.hp.hop_length) / float(self.hp.sampling_rate)).astype(np.float32)
        sym = rng.randint(1, self.hp.n_symbols, size=L).astype(np.int64)
        voiced = (d > 0).astype(np.float32)
        s_en, s_pi = rng.randn(L).astype(np.float32) * voiced, rng.randn(L).astype(np.float32) * voiced
        f_en = rng.uniform(0., 60., size=T).astype(np.float32)
        f_pi = np.where(rng.rand(T) < 0.3, 0., rng.randn(T) * 0.3 + 5.0).astype(np.float32)
        mel = np.clip(rng.randn(self.hp.n_mel_channels, T) * 2. - 5., np.log(1e-5), 2.).astype(np.float32)
        spk = int(rng.randint(0, self.n_speakers))
        t = torch.from_numpy
        return [t(sym), t(dur_f), t(d), t(s_en), t(s_pi), t(f_en), t(f_pi), t(mel), spk, 'synthetic', f'utt{idx:06d}']


def synthetic_batch(hparams, batch_size, seed=1234, **kw):
    ds = SyntheticUtterances(hparams, batch_size, seed=seed, **kw)
    return DaftExprtDataCollate(hparams)([ds[i] for i in range(batch_size)])


def synthetic_inference_batch(hparams, batch_size, seed=1234, l_range=(40, 160), t_ref_range=(250, 1000), mean_symbol_s=0.08):
    ''' collated inputs of `DaftExprt.inference` for BASELINE configs[3]: L ~ U{l_range}, sorted descending like
        `generate.collate_tensors`; reference prosody T_ref ~ U{t_ref_range} with the statistics of `SyntheticUtterances`;
        energy factors 1, pitch shift 0 ('add'); duration factors scaled so that a predictor centred on `mean_symbol_s`
        seconds per symbol keeps every utterance below ~1000 frames.  Returns the 10-tuple of CPU tensors. '''
    rng = np.random.RandomState(seed)
    B, n_mel = batch_size, hparams.n_mel_channels
    L = np.sort(rng.randint(l_range[0], l_range[1] + 1, size=B))[::-1].copy()
    Tr = rng.randint(t_ref_range[0], t_ref_range[1] + 1, size=B)
    Lm, Tm = int(L.max()), int(Tr.max())
    symbols = torch.zeros(B, Lm, dtype=torch.long)
    dur_f = torch.ones(B, Lm)
    e_ref, p_ref, m_ref = torch.zeros(B, Tm), torch.zeros(B, Tm), torch.zeros(B, n_mel, Tm)
    fps = float(hparams.sampling_rate) / float(hparams.hop_length)
    for b in range(B):
        symbols[b, :L[b]] = torch.from_numpy(rng.randint(1, hparams.n_symbols, size=L[b]))
        dur_f[b, :L[b]] = min(1., 1000. / (L[b] * mean_symbol_s * fps * 1.15))
        e_ref[b, :Tr[b]] = torch.from_numpy(rng.uniform(0, 60, size=Tr[b]).astype(np.float32))
        p_ref[b, :Tr[b]] = torch.from_numpy(np.where(rng.rand(Tr[b]) < 0.3, 0., rng.randn(Tr[b]) * 0.3 + 5.).astype(np.float32))
        m_ref[b, :, :Tr[b]] = torch.from_numpy(np.clip(rng.randn(n_mel, Tr[b]) * 2 - 5, np.log(1e-5), 2.).astype(np.float32))
    return (symbols, dur_f, torch.ones(B, Lm), torch.zeros(B, Lm), torch.from_numpy(L), e_ref, p_ref, m_ref,
            torch.from_numpy(Tr), torch.from_numpy(rng.randint(0, max(1, hparams.n_speakers - 1), size=B)))


def centre_duration_head(model, mean_symbol_s=0.08):
    ''' random-init weights predict arbitrary durations (utterances of > 2000 frames): scale the duration row of
        the predictor's projection down and centre its bias on `mean_symbol_s` so that synthetic synthesis runs have
        realistic lengths.  Bench / test helper only. '''
    with torch.no_grad():
        model._P['prosody_predictor.projection.linear_layer.weight'][0].mul_(0.05)
        model._P['prosody_predictor.projection.linear_layer.bias'].copy_(
            torch.tensor([mean_symbol_s, 0., 0.], device=model._flat.device))
        model.mark_updated()


# ------------------------------------------------------------------------------------------------
# grouped micro-batches: the reference's `accumulation_steps` micro-batches of one optimizer step as ONE padded batch
# ------------------------------------------------------------------------------------------------
class GroupedBatch(tuple):
    ''' (inputs, targets) of `accum` micro-batches concatenated along the batch axis and padded to the group's (L_max, T_max),
        plus `bounds` = (skip_in, nmax_in, skip_out, nmax_out), int64 (B_total,) tensors on the inputs' device:
        nmax_* = padded length of the utterance's OWN micro-batch -- rows at or past it do not exist for that utterance in the
        reference's step (a hard sequence end: zero on read, never written), so the kernels treat them as dead rows;
        skip_* = max(0, min(length, nmax - 2)) = what the kernels' `skip_lengths` argument has to be for rows >= min(length + 2,
        nmax) to count as dead.  With these the ONE pass over the group is the reference's accumulation of the micro-batches
        (`train.py:379-401`): its loss terms are per-utterance means averaged over the batch, so the mean over the group equals
        the sum over micro-batches of (micro-batch mean / accum). '''
    def __new__(cls, inputs, targets, bounds, accum, sizes):
        self = super().__new__(cls, (inputs, targets))
        self.inputs, self.targets, self.bounds, self.accum, self.sizes = inputs, targets, bounds, accum, sizes
        return self


def _pad_to(t, dim, size):
    if t.shape[dim] == size:
        return t
    shape = list(t.shape)
    shape[dim] = size
    out = t.new_zeros(shape)
    out.narrow(dim, 0, t.shape[dim]).copy_(t)
    return out


def group_micro_batches(micro_batches):
    ''' [(inputs 11-tuple, targets 5-tuple)] (host or device tensors, `parse_batch` order) -> GroupedBatch.  Plain torch copies:
        data staging like `parse_batch`, run once per optimizer step (or once per resident batch) outside the kernels' path. '''
    ins = [mb[0] for mb in micro_batches]
    accum = len(ins)
    Lg = max(i[0].shape[1] for i in ins)
    Tg = max(i[8].shape[2] for i in ins)
    cat = lambda idx, dim, size: torch.cat([_pad_to(i[idx], dim, size) for i in ins], 0).contiguous()
    inputs = (cat(0, 1, Lg), cat(1, 1, Lg), cat(2, 1, Lg), cat(3, 1, Lg), cat(4, 1, Lg), torch.cat([i[5] for i in ins]),
              cat(6, 1, Tg), cat(7, 1, Tg), cat(8, 2, Tg), torch.cat([i[9] for i in ins]), torch.cat([i[10] for i in ins]))
    # targets: parse_batch hands out views of the inputs (durations_float, symbols_energy, symbols_pitch, mel_specs, speaker_ids) -- then the
    # group's targets are the group's inputs; a caller's own target tensors are concatenated like the inputs
    tg = [mb[1] for mb in micro_batches]
    TIDX = (1, 3, 4, 8, 10)
    if all(t is None or all(x.data_ptr() == i[k].data_ptr() and x.shape == i[k].shape for x, k in zip(t, TIDX)) for t, i in zip(tg, ins)):
        targets = tuple(inputs[k] for k in TIDX)
    else:
        assert all(t is not None for t in tg), 'group_micro_batches: targets given for some micro-batches only'
        tcat = lambda j, dim, size: torch.cat([_pad_to(t[j], dim, size) for t in tg], 0).contiguous()
        targets = (tcat(0, 1, Lg), tcat(1, 1, Lg), tcat(2, 1, Lg), tcat(3, 2, Tg), torch.cat([t[4] for t in tg]))
    dev = inputs[5].device
    nmax_in = torch.cat([torch.full((i[0].shape[0],), i[0].shape[1], dtype=torch.long, device=dev) for i in ins])
    nmax_out = torch.cat([torch.full((i[8].shape[0],), i[8].shape[2], dtype=torch.long, device=dev) for i in ins])
    skip_in = torch.clamp(torch.minimum(inputs[5], nmax_in - 2), min=0)
    skip_out = torch.clamp(torch.minimum(inputs[9], nmax_out - 2), min=0)
    return GroupedBatch(inputs, targets, (skip_in, nmax_in, skip_out, nmax_out), accum, [i[0].shape[0] for i in ins])


def group_host_batches(batches):
    ''' the same on collate outputs (13-tuples of host tensors): returns (13-tuple of the group, (nmax_in, nmax_out) host tensors, sizes);
        `DaftExprt.parse_batch` and the clamp `skip = max(0, min(length, nmax - 2))` in `train()` finish the job on the device.
        The caller checks `len(set(sizes)) == 1` (ragged micro-batches run one pass each, see `Trainer._grouped`) '''
    ins = [(b[:11], None) for b in batches]
    g = group_micro_batches(ins)
    dirs = [d for b in batches for d in b[11]]
    files = [f for b in batches for f in b[12]]
    return tuple(g.inputs) + (dirs, files), (g.bounds[1], g.bounds[3]), g.sizes
