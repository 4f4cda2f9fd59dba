"""The random stream of the reference's seeded searches (rand 0.9 `StdRng::seed_from_u64` + `random_range(0..d)`; call sites:
tensorci2.rs:1653-1657, globalpivot.rs:174-180, adaptive_interpolation.rs:164,472-480, treetci globalpivot.rs:118-122, aci
global_guard.rs:71-74).  The crates are third party and absent from /root/reference, so the pins are the PUBLISHED vectors of the
cipher underneath (`StdRng` = ChaCha12): RFC 8439 section 2.3.2 (the 20-round block function) and the all-zero-key key streams of the
20- and 12-round variants (draft-strombergson-chacha-test-vectors TC1; rand_chacha's own tests use the same zero-key vector).  Three
independent implementations must agree on everything else: this file's numpy one, the oracle's (oracle/t4a_oracle_rng.hpp) and the
product's (tensor4all-rs_amd/csrc/stdrng.hpp, through the C ABI; host-only, runs without a GPU).  Seed expansion (PCG32) and range
sampling (widening multiply + one conditional extra draw) are restated from the crates' published sources: no fixture of the
reference fixes a seed -> point mapping, so those two steps stay 'parity unpinned' against the Rust binary."""
import numpy as np
import pytest

import oracle_binding as ob
import t4a_amd

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(v, c):
    return ((v << c) & M32) | (v >> (32 - c))


def _quarter(x, a, b, c, d):
    x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
    x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
    x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
    x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)


def py_chacha_block(key_words, counter, stream, rounds):
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + [int(k) for k in key_words] + [
        counter & M32, (counter >> 32) & M32, stream & M32, (stream >> 32) & M32]
    x = list(init)
    for _ in range(rounds // 2):
        _quarter(x, 0, 4, 8, 12); _quarter(x, 1, 5, 9, 13); _quarter(x, 2, 6, 10, 14); _quarter(x, 3, 7, 11, 15)
        _quarter(x, 0, 5, 10, 15); _quarter(x, 1, 6, 11, 12); _quarter(x, 2, 7, 8, 13); _quarter(x, 3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, init)]


class PyStdRng:
    def __init__(self, seed):
        state = seed & M64
        self.key = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & M64
            xs = (((state >> 18) ^ state) >> 27) & M32
            rot = state >> 59
            self.key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32)
        self.counter = 0
        self.buf = []
        self.index = 64

    def _refill(self, index):
        self.buf = []
        for b in range(4):
            self.buf += py_chacha_block(self.key, self.counter + b, 0, 12)
        self.counter += 4
        self.index = index

    def next_u32(self):
        if self.index >= 64:
            self._refill(0)
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self):
        if self.index < 63:
            v = (self.buf[self.index + 1] << 32) | self.buf[self.index]
            self.index += 2
            return v
        if self.index >= 64:
            self._refill(2)
            return (self.buf[1] << 32) | self.buf[0]
        x = self.buf[63]
        self._refill(1)
        return (self.buf[0] << 32) | x

    def random_range(self, n):
        if n - 1 > M32:
            m = self.next_u64() * n
            result, lo = m >> 64, m & M64
            if lo > ((-n) & M64):
                hi2 = (self.next_u64() * n) >> 64
                if lo + hi2 > M64:
                    result += 1
            return result
        rng = n & M32
        if rng == 0:
            return self.next_u32()
        m = self.next_u32() * rng
        result, lo = m >> 32, m & M32
        if lo > ((-rng) & M32):
            hi2 = (self.next_u32() * rng) >> 32
            if lo + hi2 > M32:
                result += 1
        return result


def _words_to_bytes(words):
    return b"".join(int(w).to_bytes(4, "little") for w in words)


RFC8439_2_3_2 = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
                 0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
ZERO_KEY_CHACHA20 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                  "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
ZERO_KEY_CHACHA12 = bytes.fromhex("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                                  "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")


def _all_blocks(key_words, counter, stream, rounds):
    key_bytes = np.frombuffer(_words_to_bytes(key_words), dtype=np.uint8)
    return (py_chacha_block(key_words, counter, stream, rounds),
            [int(v) for v in ob.chacha_block(key_bytes, counter, stream, rounds)],
            [int(v) for v in t4a_amd.chacha_block(key_words, counter, stream, rounds)])


def test_block_function_rfc8439_section_2_3_2():
    # key 00 01 .. 1f, block counter 1, nonce 00 00 00 09 | 00 00 00 4a | 00 00 00 00: in the 64 + 64 bit layout of rand_chacha the
    # first nonce word is the high half of the counter, the other two are the stream id
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    for got in _all_blocks(key, 1 | (0x09000000 << 32), 0x4A000000, 20):
        assert got == RFC8439_2_3_2


def test_zero_key_streams_of_the_20_and_12_round_ciphers():
    for rounds, want in ((20, ZERO_KEY_CHACHA20), (12, ZERO_KEY_CHACHA12)):
        for got in _all_blocks([0] * 8, 0, 0, rounds):
            assert _words_to_bytes(got) == want


@pytest.mark.parametrize("seed", [0, 1, 42, 0x1234567, 2**63 + 12345, M64])
def test_three_implementations_agree_on_words_and_ranges(seed):
    # raw words incl. the 64-word buffer edge taken by next_u64 (63 words first: the pair is split over a refill)
    py = PyStdRng(seed)
    w32 = [py.next_u32() for _ in range(63)]
    w64 = [py.next_u64() for _ in range(70)]
    o32, o64 = ob.stdrng_words(seed, 63, 70)
    assert [int(v) for v in o32] == w32 and [int(v) for v in o64] == w64
    # ranges: quantics bits, small dims, a non power of two that takes the second draw now and then, u32::MAX + 1 and beyond
    dims = [2] * 70 + [3, 5, 7, 10, 100, 1000, 3 * 2**30, 2**32 - 1, 2**32, 2**32 + 1, 2**40 + 12345, 2**63 + 1] * 12
    py = PyStdRng(seed)
    want = [py.random_range(d) for d in dims]
    assert all(0 <= v < d for v, d in zip(want, dims))
    assert [int(v) for v in ob.stdrng_sample(seed, dims)] == want
    assert [int(v) for v in t4a_amd.stdrng_sample(seed, dims)] == want


def test_quantics_draw_is_the_top_bit_of_the_next_word():
    # d = 2: (x * 2) >> 32 — one word per site, never a second draw
    py = PyStdRng(7)
    words = [py.next_u32() for _ in range(64)]
    assert [int(v) for v in t4a_amd.stdrng_sample(7, [2] * 64)] == [w >> 31 for w in words]


def test_second_draw_of_the_range_sampler_is_exercised():
    # range 3 * 2^30: lo > 2^32 - range happens for about three draws in four, so both arms run
    py = PyStdRng(5)
    before = py.index
    n, extra = 200, 0
    for _ in range(n):
        i0, c0 = py.index, py.counter
        py.random_range(3 * 2**30)
        used = (py.counter - c0) * 16 + py.index - i0
        extra += used == 2
    assert 0 < extra < n and before == 64


def test_empty_range_is_an_error():
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.stdrng_sample(1, [2, 0, 2])
    assert e.value.code == t4a_amd.INVALID_ARGUMENT


# ------------------------------------------------------------------------------------------------
# The reference's two other seeded streams, restated from their published algorithms in
# tensor4all-rs_amd/csrc/smallrng.hpp (product) and oracle/t4a_oracle_rng2.hpp (oracle), third implementation below:
#   TreeTCI proposers (tensor4all-treetci/src/proposer.rs:344-409): std DefaultHasher (SipHash-1-3, zero key) -> rand 0.9 SmallRng
#   (xoshiro256++, SplitMix64 seed expansion) -> random_range / shuffle (IncreasingUniform);
#   ACI initial guess (tensor4all-aci/src/random_tt.rs:31,143-150): ChaCha8Rng + rand_distr StandardNormal (256-layer ziggurat).
# Published pins: the SipHash paper's 2-4 vector, CPython 3.10's zero-key SipHash-2-4 (hash(bytes) under PYTHONHASHSEED=0), the
# xoshiro256++ reference outputs for the state (1, 2, 3, 4), rand's own seed_from_u64(0) vector, the zero-key ChaCha8 key stream.
# Unpinned against the Rust binary (no toolchain, no fixture): the byte layout `Hash` feeds the hasher, the seed -> candidate mapping,
# the ziggurat tables' last digit and libm's ln / exp.
# ------------------------------------------------------------------------------------------------
def _rotl64(v, c):
    return ((v << c) & M64) | (v >> (64 - c))


def py_siphash(msg, k0=0, k1=0, c=1, d=3):
    v0, v1 = k0 ^ 0x736F6D6570736575, k1 ^ 0x646F72616E646F6D
    v2, v3 = k0 ^ 0x6C7967656E657261, k1 ^ 0x7465646279746573

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & M64; v1 = _rotl64(v1, 13); v1 ^= v0; v0 = _rotl64(v0, 32)
        v2 = (v2 + v3) & M64; v3 = _rotl64(v3, 16); v3 ^= v2
        v0 = (v0 + v3) & M64; v3 = _rotl64(v3, 21); v3 ^= v0
        v2 = (v2 + v1) & M64; v1 = _rotl64(v1, 17); v1 ^= v2; v2 = _rotl64(v2, 32)
        return v0, v1, v2, v3
    n = len(msg)
    for w in range(n // 8):
        m = int.from_bytes(msg[8 * w:8 * w + 8], "little")
        v3 ^= m
        for _ in range(c):
            v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    last = int.from_bytes(msg[8 * (n // 8):] + b"\x00" * (7 - n % 8), "little") | ((n & 0xFF) << 56)
    v3 ^= last
    for _ in range(c):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    v0 ^= last
    v2 ^= 0xFF
    for _ in range(d):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return v0 ^ v1 ^ v2 ^ v3


class PySmallRng:
    def __init__(self, seed=None, state=None):
        if state is not None:
            self.s = list(state)
            return
        self.s, x = [], seed & M64
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & M64
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
            self.s.append(z ^ (z >> 31))

    def next_u64(self):
        s = self.s
        out = (_rotl64((s[0] + s[3]) & M64, 23) + s[0]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = _rotl64(s[3], 45)
        return out

    def next_u32(self):
        return self.next_u64() >> 32

    def below_u32(self, rng):
        if rng == 0:
            return self.next_u32()
        m = self.next_u32() * rng
        result, lo = m >> 32, m & M32
        if lo > ((-rng) & M32):
            if lo + ((self.next_u32() * rng) >> 32) > M32:
                result += 1
        return result

    def random_range(self, n):
        if n - 1 > M32:
            m = self.next_u64() * n
            result, lo = m >> 64, m & M64
            if lo > ((-n) & M64):
                if lo + ((self.next_u64() * n) >> 64) > M64:
                    result += 1
            return result
        return self.below_u32(n & M32)

    def shuffle(self, v):  # rand 0.9 partial_shuffle(len) through IncreasingUniform
        if len(v) <= 1:
            return
        n, chunk, remaining = 0, 0, 1
        for i in range(len(v)):
            nxt = n + 1
            if remaining > 0:
                nrem = remaining - 1
            else:
                bound, cur = nxt, nxt + 1
                while bound * cur <= M32:
                    bound *= cur
                    cur += 1
                chunk = self.below_u32(bound & M32)
                nrem = cur - nxt - 1
            if nrem == 0:
                idx = chunk
            else:
                idx, chunk = chunk % nxt, chunk // nxt
            remaining, n = nrem, nxt
            v[i], v[idx] = v[idx], v[i]


ZERO_KEY_CHACHA8 = bytes.fromhex("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
                                 "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")


def test_siphash_published_vectors():
    # SipHash-2-4, key 00 01 .. 0f, message 00 01 .. 0e (Aumasson & Bernstein 2012, appendix A)
    k0 = int.from_bytes(bytes(range(8)), "little")
    k1 = int.from_bytes(bytes(range(8, 16)), "little")
    msg = bytes(range(15))
    for f in (py_siphash, ob.siphash, t4a_amd.siphash):
        assert f(msg, k0, k1, 2, 4) == 0xA129CA6149BE45E5
    # CPython's bytes hash is the zero-key SipHash of the bytes when hash randomisation is off (3.10: 2-4, from 3.11: 1-3)
    import subprocess, sys
    msgs = [b"", b"a", b"tensor4all", bytes(range(7)), bytes(range(8)), bytes(range(9)), bytes(range(63)), b"simple\xff", b"x" * 200]
    code = "import sys; print([hash(m) & 0xFFFFFFFFFFFFFFFF for m in %r]); print(sys.hash_info.algorithm)" % (msgs,)
    out = subprocess.run([sys.executable, "-c", code], env={"PYTHONHASHSEED": "0"}, capture_output=True, text=True, check=True).stdout.splitlines()
    algo = out[1].strip()
    rounds = {"siphash24": (2, 4), "siphash13": (1, 3)}.get(algo)
    if rounds is None:
        pytest.skip(f"this interpreter hashes with {algo}")
    want = eval(out[0])
    for m, w in zip(msgs, want):
        if m == b"":
            continue  # (CPython returns 0 for the empty string without hashing)
        for f in (py_siphash, ob.siphash, t4a_amd.siphash):
            assert f(m, 0, 0, *rounds) == w, (m, algo)
    # 1-3 is the same code with c = 1, d = 3: the three implementations agree on it
    for m in msgs:
        assert py_siphash(m) == ob.siphash(m) == t4a_amd.siphash(m)


def test_xoshiro256plusplus_published_vectors():
    # the reference implementation's outputs for the state (1, 2, 3, 4) (rand's own test of Xoshiro256PlusPlus)
    want = [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162, 14011001112246962877,
            12406186145184390807, 15849039046786891736, 10450023813501588000]
    py = PySmallRng(state=[1, 2, 3, 4])
    assert [py.next_u64() for _ in range(10)] == want
    assert ob.smallrng_words(0, 10, state=[1, 2, 3, 4]) == want
    assert t4a_amd.smallrng_words(0, 10, state=[1, 2, 3, 4]) == want
    # seed_from_u64(0): SplitMix64 expansion (rand's test `test_xoshiro256plusplus` / SmallRng seed_from_u64(0))
    want0 = [5987356902031041503, 7051070477665621255, 6633766593972829180, 211316841551650330, 9136120204379184874, 379361710973160858,
             15813423377499357806, 15596884590815070553, 5439680534584881407, 1369371744833522710]
    py = PySmallRng(0)
    assert [py.next_u64() for _ in range(10)] == want0
    assert ob.smallrng_words(0, 10) == want0 and t4a_amd.smallrng_words(0, 10) == want0


def test_chacha8_zero_key_stream():
    got = py_chacha_block([0] * 8, 0, 0, 8)
    assert _words_to_bytes(got) == ZERO_KEY_CHACHA8
    assert _words_to_bytes(ob.chacha8_block([0] * 8, 0)) == ZERO_KEY_CHACHA8
    assert _words_to_bytes(t4a_amd.chacha_block([0] * 8, 0, 0, 8)) == ZERO_KEY_CHACHA8


@pytest.mark.parametrize("seed", [0, 1, 7, 0xDEADBEEF, M64])
def test_smallrng_three_implementations_agree(seed):
    dims = [2] * 40 + [3, 4, 10, 17, 1000, 3 * 2**30, 2**32 - 1, 2**32, 2**32 + 1, 2**50 + 3] * 10
    py = PySmallRng(seed)
    want = [py.random_range(d) for d in dims]
    assert all(0 <= v < d for v, d in zip(want, dims))
    assert ob.smallrng_sample(seed, dims) == want and t4a_amd.smallrng_sample(seed, dims) == want
    for n in (1, 2, 3, 12, 13, 14, 40, 257):
        py = PySmallRng(seed)
        v = list(range(n))
        py.shuffle(v)
        assert sorted(v) == list(range(n))
        assert ob.smallrng_shuffle(seed, n) == v and t4a_amd.smallrng_shuffle(seed, n) == v


def test_tree_edge_seed_is_siphash13_of_the_hash_byte_stream():
    for (seed, tag, u, v, hl, ni, nj) in [(0, "simple", 0, 1, 0, 1, 1), (42, "truncated_default", 3, 1, 2, 5, 7), (M64, "simple", 6, 4, 9, 0, 3)]:
        msg = seed.to_bytes(8, "little") + tag.encode() + b"\xff"
        for x in (min(u, v), max(u, v), hl, ni, nj):
            msg += x.to_bytes(8, "little")
        want = py_siphash(msg)
        assert ob.tree_edge_seed(seed, tag, u, v, hl, ni, nj) == want
        assert t4a_amd.tree_edge_seed(seed, tag, u, v, hl, ni, nj) == want


def test_chacha8_standard_normal_streams_agree_and_are_normal():
    for seed in (0, 1, 12345):
        a, wa = ob.chacha8_standard_normal(seed, 4000, 70)
        b, wb = t4a_amd.chacha8_standard_normal(seed, 4000, 70)
        assert wa == wb
        assert np.array_equal(a, b)
    # the key stream is ChaCha8 on the PCG32-expanded seed (same expansion as StdRng)
    py = PyStdRng(12345)
    words = []
    for blk in range(2):
        words += py_chacha_block(py.key, blk, 0, 8)
    assert wb[:32] == words[:32]
    # distribution: moments and tail mass of 200 000 draws (a ziggurat with a wrong table or a wrong tail is far outside these)
    x, _ = t4a_amd.chacha8_standard_normal(2024, 200000)
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1.0) < 0.01
    assert abs((np.abs(x) > 1.959964).mean() - 0.05) < 0.003
    assert abs((np.abs(x) > 3.654152885361009).mean() - 2.58e-4) < 1.5e-4  # the tail branch is taken and has the right mass
    assert abs(((x ** 4).mean()) - 3.0) < 0.08
