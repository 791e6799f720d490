/* MurmurHash3 x64 128-bit (Austin Appleby's published algorithm, public domain), restated for the attribute-summary
 * generator: the reference hashes a node's sorted predicate set with mmh3.hash128(...) -- the `mmh3` C extension,
 * absent from this image -- at /root/reference/graphs/createAttributeSum.py:25,30.  mmh3.hash128(key) (seed 0,
 * x64arch, unsigned) is the 16 output bytes read as one little-endian integer: out[0] | out[1] << 64.
 * Pinned by the reference's own data: the node ids in graphs/TEST/attr/{sum,map}/ are such hashes (tests/test_summaries.py).
 * Host code, plain C (gcc), part of librgcn_host.so. */
#include <stdint.h>
#include <string.h>

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t* p, int n) { /* n <= 8 bytes, little endian */
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

void rgcn_murmur3_x64_128(const void* key, int64_t len, uint32_t seed, uint64_t out[2]) {
    const uint8_t* data = (const uint8_t*)key;
    const int64_t nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (int64_t i = 0; i < nblocks; ++i) {
        uint64_t k1 = load_le64(data + 16 * i, 8), k2 = load_le64(data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t* tail = data + nblocks * 16;
    const int t = (int)(len & 15);
    if (t > 8) {
        uint64_t k2 = load_le64(tail + 8, t - 8);
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (t > 0) {
        uint64_t k1 = load_le64(tail, t > 8 ? 8 : t);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1;
    out[1] = h2;
}
