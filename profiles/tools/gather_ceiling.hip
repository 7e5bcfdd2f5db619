// How fast can this GPU gather scattered records?  Every lane loads `per_lane` pseudo-random records of a
// `footprint`-byte buffer, independent of each other (no pointer chasing), at full occupancy.  Patterns:
//   0  16 bytes of a random 64-byte line                       3  all 128 bytes of a random 128-byte line
//   1  48 bytes (3 loads) of a random 64-byte line              4  32 bytes of a random 32-byte record
//   2  both 64-byte halves (16 bytes each) of a random 128-byte line
//   5 / 6 / 7  16 bytes of each 64-byte line of a random aligned block of 256 / 512 / 1024 bytes
// Prints records/s and the bytes/s that are whole 64-byte lines touched.
// build: hipcc -O3 --offload-arch=gfx950 -o gather_ceiling gather_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int PATTERN>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ buf, uint32_t records, uint32_t per_lane,
                                                uint32_t *__restrict__ sink)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_lane; i += 4) {
        uint4 v[4][8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t r = mix(lane * 0x9e3779b9u + i + j) % records;
            if (PATTERN == 0) { v[j][0] = buf[(size_t)r * 4]; }
            if (PATTERN == 1) { v[j][0] = buf[(size_t)r * 4]; v[j][1] = buf[(size_t)r * 4 + 1]; v[j][2] = buf[(size_t)r * 4 + 2]; }
            if (PATTERN == 2) { v[j][0] = buf[(size_t)r * 8]; v[j][1] = buf[(size_t)r * 8 + 4]; }
            if (PATTERN == 3) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[j][k] = buf[(size_t)r * 8 + k];
            }
            if (PATTERN == 4) { v[j][0] = buf[(size_t)r * 2]; v[j][1] = buf[(size_t)r * 2 + 1]; }
            if (PATTERN >= 5) {
                constexpr int LINES = PATTERN == 5 ? 4 : PATTERN == 6 ? 8 : 16;
                uint4 acc4 = buf[(size_t)r * LINES * 4];
#pragma unroll
                for (int k = 1; k < LINES; k++) {
                    const uint4 x = buf[((size_t)r * LINES + k) * 4];
                    acc4.x ^= x.x; acc4.w += x.w;
                }
                v[j][0] = acc4;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int nk = (PATTERN == 0 || PATTERN >= 5) ? 1 : PATTERN == 1 ? 3 : PATTERN == 3 ? 8 : 2;
#pragma unroll
            for (int k = 0; k < nk; k++) acc += v[j][k].x ^ v[j][k].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The streaming ceiling beside it: every lane reads (or copies) consecutive 16-byte words, grid-stride.
__global__ __launch_bounds__(256) void k_stream_read(const uint4 *__restrict__ buf, size_t words, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = buf[i];
        acc += v.x ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

int main(int argc, char **argv)
{
    const size_t footprint = (argc > 1 ? (size_t)atof(argv[1]) : 1536) << 20;      // MiB
    const uint32_t per_lane = argc > 2 ? (uint32_t)atoi(argv[2]) : 256;
    const int waves_per_simd = argc > 3 ? atoi(argv[3]) : 5;
    uint4 *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, footprint));
    CHECK(hipMemset(buf, 1, footprint));
    CHECK(hipMalloc(&sink, 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * waves_per_simd;       // 256 threads = 4 waves = one per SIMD
    const size_t rec_bytes[8] = {64, 64, 128, 128, 32, 256, 512, 1024};
    const size_t line_bytes[8] = {64, 64, 128, 128, 32, 256, 512, 1024};
    for (int p = 0; p < 8; p++) {
        const uint32_t records = (uint32_t)(footprint / rec_bytes[p]);
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(a, 0));
            switch (p) {
            case 0: k_gather<0><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 1: k_gather<1><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 2: k_gather<2><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 3: k_gather<3><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 4: k_gather<4><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 5: k_gather<5><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 6: k_gather<6><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            case 7: k_gather<7><<<blocks, 256>>>(buf, records, per_lane, sink); break;
            }
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double n = (double)blocks * 256 * per_lane;
        printf("pattern %d  footprint %zu MiB  waves/SIMD %d: %.1f G records/s, %.2f TB/s of %zu-byte records, %.3f ms\n", p,
               footprint >> 20, waves_per_simd, n / ms / 1e6, n * line_bytes[p] / ms / 1e9, line_bytes[p], ms);
    }
    {
        const size_t words = footprint / 16;
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float ms;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a, 0));
            k_stream_read<<<prop.multiProcessorCount * 8, 256>>>(buf, words, sink);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
        }
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("stream read  footprint %zu MiB: %.2f TB/s, %.3f ms\n", footprint >> 20, footprint / ms / 1e9, ms);
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a, 0));
            k_stream_copy<<<prop.multiProcessorCount * 8, 256>>>(buf, buf + words / 2, words / 2);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
        }
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("stream copy  %zu MiB -> %zu MiB: %.2f TB/s read + written, %.3f ms\n", footprint >> 21, footprint >> 21,
               footprint / ms / 1e9, ms);
    }
    return 0;
}
