"""Vectorised seeding of numpy-compatible PCG64 streams (host side of the on-device noise generator).

``numpy.random.Generator(PCG64(seed))`` turns an integer seed into a 128-bit LCG state and increment through
``SeedSequence`` (numpy/random/bit_generator.pyx: 32-bit hash-mix pool of 4 words, then ``generate_state``) and
``pcg64_set_seed`` (numpy/random/src/pcg64: state = 0; step; state += initstate; step, with inc = (initseq << 1) | 1).
The Monte-Carlo driver needs half a million of these per sweep (noise.py:59,70: one generator per feature, mixture
component and selector of every trial); constructing them one by one through numpy costs ~10 s per 65 536 trials, so the
same arithmetic is done here on whole arrays.  ``tests/test_host_logic.py`` checks it against numpy's own states.
"""
import numpy as np

_INIT_A, _MULT_A, _INIT_B, _MULT_B = 0x43b0d7e5, 0x931e8875, 0x8b51f9dd, 0x58f38ded
_MIX_L, _MIX_R, _XS = 0xca01f9dd, 0x4973f715, 16
_M32 = np.uint64(0xffffffff)
PCG_MULT_HI, PCG_MULT_LO = 0x2360ED051FC65DA4, 0x4385DF649FCCF645        # PCG_DEFAULT_MULTIPLIER_128


def _u(x):
    return np.asarray(x, dtype=np.uint64)


def _seedseq_words(seeds):
    """generate_state(4, uint64) as 8 uint32 words per seed (seeds < 2**64: one or two entropy words)."""
    seeds = _u(seeds)
    e0, e1 = seeds & _M32, seeds >> np.uint64(32)
    two = e1 != 0                                                  # seeds >= 2**32 carry a second entropy word
    hc = np.full(seeds.shape, _INIT_A, dtype=np.uint64)

    def hashmix(v):
        nonlocal hc
        v = v ^ hc
        hc = (hc * np.uint64(_MULT_A)) & _M32
        v = (v * hc) & _M32
        return v ^ (v >> np.uint64(_XS))

    def mix(x, y):
        r = (np.uint64(_MIX_L) * x - np.uint64(_MIX_R) * y) & _M32
        return r ^ (r >> np.uint64(_XS))

    zero = np.zeros_like(seeds)
    pool = [hashmix(e0), hashmix(np.where(two, e1, zero)), hashmix(zero), hashmix(zero)]
    for src in range(4):
        for dst in range(4):
            if src != dst:
                pool[dst] = mix(pool[dst], hashmix(pool[src]))
    hcb = np.full(seeds.shape, _INIT_B, dtype=np.uint64)
    out = []
    for i in range(8):
        d = pool[i % 4] ^ hcb
        hcb = (hcb * np.uint64(_MULT_B)) & _M32
        d = (d * hcb) & _M32
        out.append(d ^ (d >> np.uint64(_XS)))
    return out


def _mul64(a, b):
    """(hi, lo) of the 128-bit product of uint64 arrays a, b."""
    a0, a1, b0, b1 = a & _M32, a >> np.uint64(32), b & _M32, b >> np.uint64(32)
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = (p00 >> np.uint64(32)) + (p01 & _M32) + (p10 & _M32)
    lo = (p00 & _M32) | ((mid & _M32) << np.uint64(32))
    hi = p11 + (p01 >> np.uint64(32)) + (p10 >> np.uint64(32)) + (mid >> np.uint64(32))
    return hi, lo


def _add128(ah, al, bh, bl):
    lo = al + bl
    return ah + bh + (lo < al).astype(np.uint64), lo


def _mul128_const(ah, al):
    """(ah:al) * PCG multiplier mod 2**128."""
    mh, ml = np.uint64(PCG_MULT_HI), np.uint64(PCG_MULT_LO)
    hi, lo = _mul64(al, np.full_like(al, ml))
    return hi + al * mh + ah * ml, lo


def pcg64_states(seeds):
    """seeds: array of non-negative ints < 2**64 -> uint64 array (..., 4) = (state_hi, state_lo, inc_hi, inc_lo), exactly
    ``PCG64(seed).state['state']`` for every seed."""
    with np.errstate(over='ignore'):
        w = _seedseq_words(seeds)
        u = [w[2 * i] | (w[2 * i + 1] << np.uint64(32)) for i in range(4)]
        init_hi, init_lo, seq_hi, seq_lo = u[0], u[1], u[2], u[3]
        inc_hi = (seq_hi << np.uint64(1)) | (seq_lo >> np.uint64(63))
        inc_lo = (seq_lo << np.uint64(1)) | np.uint64(1)
        sh, sl = _add128(inc_hi, inc_lo, init_hi, init_lo)        # (0 * mult + inc) + initstate
        sh, sl = _mul128_const(sh, sl)
        sh, sl = _add128(sh, sl, inc_hi, inc_lo)
    return np.stack([sh, sl, inc_hi, inc_lo], axis=-1)
