"""Jump-ahead for mt19937 (host side): which XOR of 19 937 consecutive outputs equals the output J steps later.

The mt19937 recurrence is linear over GF(2); its untempered word stream x_0, x_1, ... is annihilated by the generator's
characteristic polynomial phi(x) (degree 19 937, 135 terms).  With g(x) = x^J mod phi(x) = sum_i g_i x^i,

        x_{k+J} = XOR_{i : g_i = 1} x_{k+i}          for every k,

so 624 consecutive words J steps ahead — a complete generator state — are 624 sliding XORs over the next 19 937 + 623 words of
the stream.  `csrc/torch_rng.hip: vn_mt19937_jump_kernel` evaluates that with one workgroup per target offset, which turns the
serial walk through torch's CPU stream (`rng="torch_device"`) into independent 2 M-word chunks, one per compute unit.

g is computed here with Python integers as GF(2)[x] bit vectors (squaring = bit spreading, reduction by the sparse phi);
~0.1-0.16 s per offset the first time an offset is seen (a whole-call plan of torch_rng.draw_units needs one per sampling step + one
per chunk: 34 for the coarse stage at B = 8), cached in memory and under vampnet_amd/.cache/.  PHI_EXPONENTS was obtained with Berlekamp-Massey on the generator's own output
(tests/test_host_logic.py::test_mt19937_characteristic_polynomial recomputes it).
"""
import functools
import os

import numpy as np

MT_DEGREE = 19937
PHI_EXPONENTS = (
    0, 1189, 1416, 1585, 1643, 1870, 2493, 2773, 3000, 3227, 3454, 3681, 3908, 4135, 4362, 4753, 5661, 6337, 6569, 7129, 7477,
    7525, 7583, 7752, 7979, 8206, 9505, 9901, 9969, 10128, 10693, 10761, 10920, 11089, 11147, 11157, 11215, 11321, 11374, 11384,
    11485, 11611, 11712, 11717, 11838, 11881, 11944, 11997, 12277, 12335, 12393, 12504, 12509, 12620, 12673, 12731, 12736, 12789,
    12905, 12958, 12963, 13137, 13185, 13190, 13243, 13301, 13412, 13528, 13533, 13639, 13697, 13760, 13813, 13866, 14093, 14151,
    14209, 14320, 14325, 14436, 14547, 14552, 14605, 14721, 14774, 14779, 14953, 15001, 15006, 15059, 15117, 15228, 15344, 15349,
    15455, 15513, 15576, 15629, 15682, 15909, 15967, 16025, 16136, 16141, 16252, 16363, 16368, 16421, 16537, 16590, 16595, 16817,
    16822, 16875, 16933, 17044, 17160, 17271, 17329, 17445, 17498, 17725, 17783, 17841, 17952, 18068, 18179, 18237, 18406, 18633,
    18691, 18860, 19087, 19314, 19937)
PHI = sum(1 << e for e in PHI_EXPONENTS)


def _square(a: int) -> int:
    """a(x)^2 in GF(2)[x]: spread the bits (cross terms vanish in characteristic 2)."""
    return int("0".join(bin(a)[2:]), 2)


def _mod(a: int) -> int:
    while a.bit_length() > MT_DEGREE:
        a ^= PHI << (a.bit_length() - 1 - MT_DEGREE)
    return a


def poly_mul(a: int, b: int) -> int:
    """a(x) * b(x) mod phi(x)."""
    r = 0
    while b:
        low = b & -b
        r ^= a << (low.bit_length() - 1)
        b ^= low
    return _mod(r)


@functools.lru_cache(maxsize=4096)
def jump_poly(steps: int) -> int:
    """x^steps mod phi(x) as an integer bit vector (bit i = coefficient of x^i)."""
    if steps < 0:
        raise ValueError("cannot jump backwards")
    r = 1
    for bit in bin(steps)[2:]:
        r = _mod(_square(r))
        if bit == "1":
            r = _mod(r << 1)
    return r


def _cache_dir():
    """On-disk cache of jump polynomials (2 496 bytes each): VN_CACHE_DIR, else <package>/.cache.  Best effort: any I/O
    problem just means the polynomial is recomputed (~40 ms)."""
    return os.environ.get("VN_CACHE_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), ".cache")


def jump_poly_words(steps: int) -> np.ndarray:
    """jump_poly as 624 little-endian uint32 words (bits 19 937 .. 19 967 are zero) for the device kernel.  A process that
    draws a new noise shape needs ~20 of these; they only depend on `steps`, so they are kept on disk between runs."""
    path = os.path.join(_cache_dir(), f"mtjump_{int(steps)}.bin")
    try:
        raw = open(path, "rb").read()
        if len(raw) == 624 * 4:
            return np.frombuffer(raw, dtype="<u4").astype(np.uint32)
    except OSError:
        pass
    words = np.frombuffer(jump_poly(steps).to_bytes(624 * 4, "little"), dtype="<u4").astype(np.uint32)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(words.astype("<u4").tobytes())
        os.replace(tmp, path)                                   # atomic: concurrent ranks may race for the same file
    except OSError:
        pass
    return words


def berlekamp_massey_gf2(bits) -> int:
    """Connection polynomial C(x) (bit j = c_j, c_0 = 1) of the shortest LFSR generating `bits`."""
    c, b, length, m, win = 1, 1, 0, 1, 0
    for i, s in enumerate(bits):
        win = (win << 1) | int(s)
        if bin(c & win).count("1") & 1:
            t = c
            c ^= b << m
            if 2 * length <= i:
                length, b, m = i + 1 - length, t, 1
            else:
                m += 1
        else:
            m += 1
    return c, length
