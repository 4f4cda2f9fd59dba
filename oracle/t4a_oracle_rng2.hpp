// t4a_oracle_rng2.hpp — TEST INFRASTRUCTURE ONLY: CPU restatement of the two random streams of the reference that are NOT `StdRng`
// (t4a_oracle_rng.hpp has that one).  Never included, linked or loaded by anything under tensor4all-rs_amd/.
//
//   tensor4all-treetci/src/proposer.rs:360-387  rng_for_edge: std DefaultHasher (SipHash-1-3, zero key) over seed, tag, edge, history
//                                               length and the two pivot counts -> rand 0.9 SmallRng::seed_from_u64 (xoshiro256++)
//   proposer.rs:344-358 / :389-409              random_range(0..d) per site / slice.shuffle (rand 0.9 IncreasingUniform)
//   tensor4all-aci/src/random_tt.rs:31,143-150  ChaCha8Rng::seed_from_u64 + rand_distr 0.5 StandardNormal (256-layer ziggurat)
// The crates are un-vendored third-party code (Cargo.toml: rand = "0.9", rand_chacha = "0.9", rand_distr = "0.5"): restated from
// their published algorithms, written independently of the product's csrc/smallrng.hpp (byte-vector SipHash instead of a
// streaming one, an explicit IncreasingUniform object, ...) so that tests/test_cpu_stdrng.py can hold three implementations against
// each other and against the published vectors (SipHash paper 2-4 vector, CPython's zero-key SipHash-2-4, xoshiro256++ reference
// outputs for state 1, 2, 3, 4, rand's seed_from_u64(0) vector, ChaCha8 zero-key key stream).
// Parity unpinned against the Rust binary: the byte layout `Hash` feeds the hasher, the ziggurat tables' last digit, libm's ln / exp.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace t4a_oracle {

inline uint64_t rng2_rotl64(uint64_t v, unsigned c) { return (v << c) | (v >> (64 - c)); }

// SipHash-c-d of a whole message (Aumasson & Bernstein, reference formulation)
inline uint64_t siphash(const std::vector<uint8_t>& msg, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds)
{
    uint64_t v[4] = {k0 ^ 0x736f6d6570736575ull, k1 ^ 0x646f72616e646f6dull, k0 ^ 0x6c7967656e657261ull, k1 ^ 0x7465646279746573ull};
    auto sipround = [&]() {
        v[0] += v[1]; v[1] = rng2_rotl64(v[1], 13); v[1] ^= v[0]; v[0] = rng2_rotl64(v[0], 32);
        v[2] += v[3]; v[3] = rng2_rotl64(v[3], 16); v[3] ^= v[2];
        v[0] += v[3]; v[3] = rng2_rotl64(v[3], 21); v[3] ^= v[0];
        v[2] += v[1]; v[1] = rng2_rotl64(v[1], 17); v[1] ^= v[2]; v[2] = rng2_rotl64(v[2], 32);
    };
    const size_t n = msg.size(), full = n / 8;
    for (size_t w = 0; w < full; ++w) {
        uint64_t m = 0;
        for (int b = 7; b >= 0; --b) m = (m << 8) | msg[8 * w + (size_t)b];
        v[3] ^= m;
        for (int r = 0; r < c_rounds; ++r) sipround();
        v[0] ^= m;
    }
    uint64_t last = (uint64_t)(n & 0xff) << 56;
    for (size_t b = 8 * full; b < n; ++b) last |= (uint64_t)msg[b] << (8 * (b - 8 * full));
    v[3] ^= last;
    for (int r = 0; r < c_rounds; ++r) sipround();
    v[0] ^= last;
    v[2] ^= 0xff;
    for (int r = 0; r < d_rounds; ++r) sipround();
    return v[0] ^ v[1] ^ v[2] ^ v[3];
}

// what `Hash` writes: integers as 8 little-endian bytes, a str as its bytes + 0xff
struct HashBytes {
    std::vector<uint8_t> b;
    void u64(uint64_t v)
    {
        for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void str(const std::string& s)
    {
        b.insert(b.end(), s.begin(), s.end());
        b.push_back(0xff);
    }
    uint64_t default_hasher_finish() const { return siphash(b, 0, 0, 1, 3); } // DefaultHasher::new()
};

struct OracleSmallRng { // xoshiro256++
    uint64_t s[4];
    explicit OracleSmallRng(uint64_t seed) // seed_from_u64: SplitMix64
    {
        uint64_t x = seed;
        for (uint64_t& w : s) {
            x += 0x9e3779b97f4a7c15ull;
            uint64_t z = x;
            z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
            z ^= z >> 27; z *= 0x94d049bb133111ebull;
            z ^= z >> 31;
            w = z;
        }
    }
    uint64_t next_u64()
    {
        const uint64_t out = rng2_rotl64(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rng2_rotl64(s[3], 45);
        return out;
    }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    uint32_t below_u32(uint32_t range) // Canon's method on u32 (range == 0: all of u32)
    {
        if (range == 0) return next_u32();
        const uint64_t wide = (uint64_t)next_u32() * (uint64_t)range;
        uint32_t hi = (uint32_t)(wide >> 32);
        const uint32_t lo = (uint32_t)wide;
        if (lo > (uint32_t)(~range + 1u)) {
            const uint32_t hi2 = (uint32_t)(((uint64_t)next_u32() * (uint64_t)range) >> 32);
            if ((uint64_t)lo + (uint64_t)hi2 > 0xFFFFFFFFull) ++hi;
        }
        return hi;
    }
    size_t range(size_t n) // random_range(0..n): u32 sampling when n fits, u64 otherwise
    {
        if ((uint64_t)n <= 0x100000000ull) return (size_t)below_u32((uint32_t)n);
        const unsigned __int128 wide = (unsigned __int128)next_u64() * (unsigned __int128)n;
        uint64_t hi = (uint64_t)(wide >> 64);
        const uint64_t lo = (uint64_t)wide;
        if (lo > (uint64_t)(~(uint64_t)n + 1ull)) {
            const uint64_t hi2 = (uint64_t)(((unsigned __int128)next_u64() * (unsigned __int128)n) >> 64);
            if (lo + hi2 < lo) ++hi;
        }
        return (size_t)hi;
    }
};

// rand 0.9 src/seq/increasing_uniform.rs
struct OracleIncreasingUniform {
    OracleSmallRng& rng;
    uint32_t n, chunk = 0;
    uint8_t chunk_remaining;
    OracleIncreasingUniform(OracleSmallRng& r, uint32_t n0) : rng(r), n(n0), chunk_remaining(n0 == 0 ? 1 : 0) {}
    static void bound_of(uint32_t m, uint32_t& product, uint8_t& count)
    {
        product = m;
        uint32_t current = m + 1;
        while ((uint64_t)product * current <= 0xFFFFFFFFull) {
            product *= current;
            ++current;
        }
        count = (uint8_t)(current - m);
    }
    size_t next_index()
    {
        const uint32_t next_n = n + 1;
        uint8_t next_remaining;
        if (chunk_remaining >= 1) {
            next_remaining = (uint8_t)(chunk_remaining - 1);
        } else {
            uint32_t bound;
            uint8_t remaining;
            bound_of(next_n, bound, remaining);
            chunk = rng.below_u32(bound);
            next_remaining = (uint8_t)(remaining - 1);
        }
        size_t result;
        if (next_remaining == 0) {
            result = chunk;
        } else {
            result = chunk % next_n;
            chunk /= next_n;
        }
        chunk_remaining = next_remaining;
        n = next_n;
        return result;
    }
};
template <class T> inline void rng2_shuffle(std::vector<T>& v, OracleSmallRng& rng) // slice.shuffle(rng)
{
    if (v.size() <= 1) return;
    OracleIncreasingUniform chooser(rng, 0);
    for (size_t i = 0; i < v.size(); ++i) std::swap(v[i], v[chooser.next_index()]);
}

struct OracleChaCha8Rng {
    uint32_t key[8];
    uint64_t counter = 0;
    std::vector<uint32_t> buf;
    size_t index = 64;
    explicit OracleChaCha8Rng(uint64_t state)
    {
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
            const unsigned rot = (unsigned)(state >> 59);
            key[i] = rot == 0 ? xs : ((xs >> rot) | (xs << (32 - rot)));
        }
    }
    static void block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16])
    {
        uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                           (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t w[16];
        std::memcpy(w, st, sizeof(w));
        auto rot = [](uint32_t v, int c) { return (v << c) | (v >> (32 - c)); };
        auto quarter = [&](int a, int b, int c, int d) {
            w[a] += w[b]; w[d] ^= w[a]; w[d] = rot(w[d], 16);
            w[c] += w[d]; w[b] ^= w[c]; w[b] = rot(w[b], 12);
            w[a] += w[b]; w[d] ^= w[a]; w[d] = rot(w[d], 8);
            w[c] += w[d]; w[b] ^= w[c]; w[b] = rot(w[b], 7);
        };
        for (int r = 0; r < rounds / 2; ++r) {
            quarter(0, 4, 8, 12); quarter(1, 5, 9, 13); quarter(2, 6, 10, 14); quarter(3, 7, 11, 15);
            quarter(0, 5, 10, 15); quarter(1, 6, 11, 12); quarter(2, 7, 8, 13); quarter(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) out[i] = w[i] + st[i];
    }
    void refill(size_t at)
    {
        buf.assign(64, 0u);
        for (int b = 0; b < 4; ++b) block(key, counter + (uint64_t)b, 8, buf.data() + 16 * b);
        counter += 4;
        index = at;
    }
    uint32_t next_u32()
    {
        if (index >= 64) refill(0);
        return buf[index++];
    }
    uint64_t next_u64()
    {
        if (index < 63) {
            const uint64_t v = ((uint64_t)buf[index + 1] << 32) | buf[index];
            index += 2;
            return v;
        }
        if (index >= 64) {
            refill(2);
            return ((uint64_t)buf[1] << 32) | buf[0];
        }
        const uint64_t lo = buf[63];
        refill(1);
        return ((uint64_t)buf[0] << 32) | lo;
    }
};

struct OracleZiggurat {
    double x[257], f[257];
    static constexpr double R = 3.654152885361008796;
    OracleZiggurat()
    {
        const double v = 4.92867323399e-3;
        x[0] = v / std::exp(-0.5 * R * R);
        x[1] = R;
        for (int i = 2; i <= 255; ++i) {
            const double p = x[i - 1];
            x[i] = std::sqrt(-2.0 * std::log(v / p + std::exp(-p * p / 2.0)));
        }
        x[256] = 0.0;
        for (int i = 0; i <= 256; ++i) f[i] = std::exp(-x[i] * x[i] / 2.0);
    }
    static double from_bits(uint64_t b)
    {
        double d;
        std::memcpy(&d, &b, 8);
        return d;
    }
    template <class G> double normal(G& g) const // rand_distr::StandardNormal
    {
        while (true) {
            const uint64_t bits = g.next_u64();
            const unsigned layer = (unsigned)(bits & 255u);
            const double u = from_bits((bits >> 12) | (1024ull << 52)) - 3.0;
            const double v = u * x[layer];
            if (std::fabs(v) < x[layer + 1]) return v;
            if (layer == 0) {
                double tx = 1.0, ty = 0.0;
                do {
                    const double a = from_bits((g.next_u64() >> 12) | (1023ull << 52)) - (1.0 - 1.1102230246251565e-16);
                    const double b = from_bits((g.next_u64() >> 12) | (1023ull << 52)) - (1.0 - 1.1102230246251565e-16);
                    tx = std::log(a) / R;
                    ty = std::log(b);
                } while (-2.0 * ty < tx * tx);
                return u < 0.0 ? tx - R : R - tx;
            }
            const double unif = (double)(g.next_u64() >> 11) / 9007199254740992.0;
            if (f[layer + 1] + (f[layer] - f[layer + 1]) * unif < std::exp(-v * v / 2.0)) return v;
        }
    }
};
inline const OracleZiggurat& oracle_ziggurat()
{
    static const OracleZiggurat z;
    return z;
}

} // namespace t4a_oracle
