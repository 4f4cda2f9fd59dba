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
