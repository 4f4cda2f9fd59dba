// t4a_oracle_rng.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README): CPU restatement of the random stream the reference draws its
// search points from.  Never included, linked or loaded by anything under tensor4all-rs_amd/.
//
// Reference call sites: `rand::rngs::StdRng::seed_from_u64(seed)` then `rng.random_range(0..d)`
//   tensor4all-tensorci/src/tensorci2.rs:1653-1657 + globalpivot.rs:174-180 (DefaultGlobalPivotFinder),
//   tensor4all-partitionedtt/src/adaptive_interpolation.rs:164,472-480, tensor4all-treetci/src/globalpivot.rs:118-122,
//   tensor4all-aci/src/global_guard.rs:71-74, tensor4all-tensorci/src/globalsearch.rs:93-99.
// The algorithm lives in third-party crates that are NOT under /root/reference (Cargo.toml:75-76 pins rand = "0.9",
// rand_chacha = "0.9"; no Cargo.lock is vendored): restated from their published definitions —
//   rand 0.9      StdRng = rand_chacha::ChaCha12Rng; usize ranges: UniformUsize -> UniformInt<u32/u64>::sample_single_inclusive
//                 (widening multiply, one conditional extra draw: "Canon's method")
//   rand_chacha   ChaCha, 12 rounds, key = seed bytes, 64-bit counter (words 12, 13) from 0, 64-bit stream id 0 (words 14, 15),
//                 key stream consumed as little-endian u32 words through rand_core::block::BlockRng (64-word buffer)
//   rand_core 0.9 SeedableRng::seed_from_u64: PCG32 stream -> seed bytes
// Pinned (tests/test_cpu_stdrng.py): the block function against RFC 8439 section 2.3.2 and the zero-key ChaCha20 / ChaCha12 vectors
// (draft-strombergson-chacha-test-vectors TC1), and this file against the product's independent implementation and a numpy one.
// Parity unpinned: seed expansion and range sampling (no fixture of the reference fixes a seed -> point mapping).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace t4a_oracle {

struct OracleStdRng {
    std::array<uint8_t, 32> seed{}; // ChaCha key bytes
    uint64_t block_counter = 0;
    std::vector<uint32_t> results;  // BlockRng buffer (4 blocks)
    size_t index = 64;

    explicit OracleStdRng(uint64_t state = 0) // SeedableRng::seed_from_u64
    {
        for (size_t chunk = 0; chunk < 8; ++chunk) {
            state = state * 6364136223846793005ull + 11634580027462260723ull; // (wrapping)
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            const unsigned rot = (unsigned)(state >> 59);
            const uint32_t x = rot == 0 ? xorshifted : ((xorshifted >> rot) | (xorshifted << (32 - rot)));
            for (size_t b = 0; b < 4; ++b) seed[4 * chunk + b] = (uint8_t)(x >> (8 * b)); // to_le_bytes
        }
        results.assign(64, 0u);
    }

    static uint32_t rotate_left(uint32_t v, unsigned n) { return (v << n) | (v >> (32 - n)); }
    static void quarter_round(std::array<uint32_t, 16>& s, size_t a, size_t b, size_t c, size_t d)
    {
        s[a] += s[b]; s[d] ^= s[a]; s[d] = rotate_left(s[d], 16);
        s[c] += s[d]; s[b] ^= s[c]; s[b] = rotate_left(s[b], 12);
        s[a] += s[b]; s[d] ^= s[a]; s[d] = rotate_left(s[d], 8);
        s[c] += s[d]; s[b] ^= s[c]; s[b] = rotate_left(s[b], 7);
    }
    static std::array<uint32_t, 16> chacha_block(const std::array<uint8_t, 32>& key, uint64_t counter, uint64_t stream, unsigned double_rounds)
    {
        std::array<uint32_t, 16> init{};
        const char sigma[17] = "expand 32-byte k";
        for (size_t w = 0; w < 4; ++w)
            init[w] = (uint32_t)(uint8_t)sigma[4 * w] | ((uint32_t)(uint8_t)sigma[4 * w + 1] << 8) | ((uint32_t)(uint8_t)sigma[4 * w + 2] << 16) |
                      ((uint32_t)(uint8_t)sigma[4 * w + 3] << 24);
        for (size_t w = 0; w < 8; ++w)
            init[4 + w] = (uint32_t)key[4 * w] | ((uint32_t)key[4 * w + 1] << 8) | ((uint32_t)key[4 * w + 2] << 16) | ((uint32_t)key[4 * w + 3] << 24);
        init[12] = (uint32_t)(counter & 0xFFFFFFFFull);
        init[13] = (uint32_t)(counter >> 32);
        init[14] = (uint32_t)(stream & 0xFFFFFFFFull);
        init[15] = (uint32_t)(stream >> 32);
        std::array<uint32_t, 16> work = init;
        for (unsigned i = 0; i < double_rounds; ++i) {
            for (size_t col = 0; col < 4; ++col) quarter_round(work, col, 4 + col, 8 + col, 12 + col);
            for (size_t dg = 0; dg < 4; ++dg) quarter_round(work, dg, 4 + (dg + 1) % 4, 8 + (dg + 2) % 4, 12 + (dg + 3) % 4);
        }
        for (size_t w = 0; w < 16; ++w) work[w] += init[w];
        return work;
    }

    void generate_and_set(size_t new_index)
    {
        for (uint64_t b = 0; b < 4; ++b) {
            const std::array<uint32_t, 16> blk = chacha_block(seed, block_counter + b, 0, 6);
            for (size_t w = 0; w < 16; ++w) results[16 * b + w] = blk[w];
        }
        block_counter += 4;
        index = new_index;
    }
    uint32_t next_u32()
    {
        if (index >= results.size()) generate_and_set(0);
        return results[index++];
    }
    uint64_t next_u64() // rand_core::block::BlockRng::next_u64
    {
        const size_t len = results.size();
        if (index < len - 1) {
            index += 2;
            return ((uint64_t)results[index - 1] << 32) | (uint64_t)results[index - 2];
        } else if (index >= len) {
            generate_and_set(2);
            return ((uint64_t)results[1] << 32) | (uint64_t)results[0];
        }
        const uint64_t x = results[len - 1];
        generate_and_set(1);
        const uint64_t y = results[0];
        return (y << 32) | x;
    }
    // rng.random_range(0..high) on usize (high > 0)
    size_t range(size_t high)
    {
        const uint64_t top = (uint64_t)high - 1; // inclusive end
        if (top > 0xFFFFFFFFull) {
            const uint64_t range = top + 1; // (high <= 2^64 - 1: cannot wrap)
            const unsigned __int128 wide = (unsigned __int128)next_u64() * (unsigned __int128)range;
            uint64_t result = (uint64_t)(wide >> 64);
            const uint64_t lo_order = (uint64_t)wide;
            if (lo_order > (uint64_t)0 - range) {
                const uint64_t new_hi_order = (uint64_t)(((unsigned __int128)next_u64() * (unsigned __int128)range) >> 64);
                const bool is_overflow = lo_order > UINT64_MAX - new_hi_order; // checked_add(...).is_none()
                result += is_overflow ? 1 : 0;
            }
            return (size_t)result;
        }
        const uint32_t range = (uint32_t)top + 1u; // wraps to 0 for the full u32 range
        if (range == 0) return (size_t)next_u32();
        const uint64_t wide = (uint64_t)next_u32() * (uint64_t)range;
        uint32_t result = (uint32_t)(wide >> 32);
        const uint32_t lo_order = (uint32_t)(wide & 0xFFFFFFFFull);
        if (lo_order > (uint32_t)0 - range) {
            const uint32_t new_hi_order = (uint32_t)(((uint64_t)next_u32() * (uint64_t)range) >> 32);
            const bool is_overflow = lo_order > UINT32_MAX - new_hi_order;
            result += is_overflow ? 1u : 0u;
        }
        return (size_t)result;
    }
};

} // namespace t4a_oracle
