"""numpy restatement of the dropout masks the HIP kernels draw (csrc/dx_common.h: dx_mix32, dx_key32, dx_keep_elem and the
attention-weight block hash), so that the CPU oracle can run the reference's training forward / backward with THE masks of a HIP
pass (tests/test_gpu_dropout_parity.py): the timed configuration -- dropout on -- compared element by element.

Test infrastructure: nothing in the package imports this.  All arithmetic is modulo 2^32 on uint64 arrays."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
M24 = np.uint64(0xFFFFFF)
CTR_MUL = np.uint64(0x9E3779B1)
M24_PRE = np.uint64(0x9E3779)
BLK_M = np.array([0xC2B2AF, 0x85EBCB, 0xA54FF5, 0x6C8E95], dtype=np.uint64)


def _u(x):
    return np.asarray(x, dtype=np.uint64)


def mix32(x):
    x = _u(x) & M32
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x85ebca6b)) & M32
    x = x ^ (x >> np.uint64(13))
    x = (x * np.uint64(0xc2b2ae35)) & M32
    return x ^ (x >> np.uint64(16))


def key32(seed, salt):
    ''' dx_key32: the 32-bit stream key of (63-bit seed, salt) '''
    seed = int(seed)
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    inner = mix32((hi + 0x9E3779B9 * (int(salt) + 1)) & 0xFFFFFFFF)
    return mix32(np.uint64(lo) ^ inner)


def drop_th8(p):
    if p <= 0.:
        return 0
    t = int(np.float32(p) * np.float32(256.) + np.float32(0.5))
    return min(max(t, 1), 255)


def inv_keep8(th8):
    return float(np.float32(256.) / np.float32(256 - th8))


def elem_keep(seed, stream, rows, N, C, p):
    ''' dx_keep_elem over a (B, N, C) activation: keep[b', n, c] for the batch rows `rows` (indices into the batch the kernel ran on).
        stream 0 = dropout in FRONT of a LayerNorm (GEMM epilogues, ln_fwd p_pre), 1 = behind it (ln_fwd p_post).
        Returns (bool array (len(rows), N, C), scale of the kept values). '''
    th8 = drop_th8(p)
    key = key32(seed, stream)
    b = _u(rows)[:, None, None]
    n = _u(np.arange(N))[None, :, None]
    c = _u(np.arange(C))[None, None, :]
    idx = ((b * np.uint64(N) + n) * np.uint64(C) + c) & M32
    h = mix32(((idx >> np.uint64(2)) * CTR_MUL + key) & M32)
    byte = (h >> ((idx & np.uint64(3)) * np.uint64(8))) & np.uint64(0xFF)
    return byte >= np.uint64(th8), inv_keep8(th8)


def attn_keep(seed, rows, H, N, p):
    ''' attention-weight dropout: keep[b', h, q, key] for the batch rows `rows`; one block hash per 4 x 4 (query, key) block,
        one row word per query of the block, one byte per key (dx_common.h) '''
    th8 = drop_th8(p)
    NB = np.uint64((N + 3) >> 2)
    q = _u(np.arange(N))[:, None]
    k = _u(np.arange(N))[None, :]
    blk = ((q >> np.uint64(2)) * NB + (k >> np.uint64(2))) * CTR_MUL
    rot = (q & np.uint64(3)) * np.uint64(8)
    mult = BLK_M[(q & np.uint64(3)).astype(np.int64)]
    shift = (k & np.uint64(3)) * np.uint64(8)
    out = np.empty((len(rows), H, N, N), dtype=bool)
    for i, b in enumerate(rows):
        for h in range(H):
            c = (blk + key32(seed, int(b) * H + h)) & M32
            c = c ^ (c >> np.uint64(16))
            c = ((c & M24) * M24_PRE) & M32
            c = c ^ (c >> np.uint64(16))
            w = ((c >> rot) | (c << (np.uint64(32) - rot))) & M32       # rotr (rot = 0: c | c << 32, masked back to c)
            w = ((w & M24) * mult) & M32
            out[i, h] = ((w >> shift) & np.uint64(0xFF)) >= np.uint64(th8)
    return out, inv_keep8(th8)
