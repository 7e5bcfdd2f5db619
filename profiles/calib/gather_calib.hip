// Calibration of FETCH_SIZE for the access pattern of the traversal kernels:
// every lane reads 16 bytes (mode 0) or 48 bytes (mode 1, three 16-byte loads,
// 48-byte stride records) at a pseudo-random position of a table far larger
// than the caches.  Known request count -> compare with the counter.
//   build: hipcc -O3 --offload-arch=gfx950 gather_calib.hip -o gather_calib
//   run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t z)
{
    z ^= z >> 16; z *= 0x85ebca6bu; z ^= z >> 13; z *= 0xc2b2ae35u; z ^= z >> 16;
    return z;
}

__global__ void k_gather16(const uint4 *__restrict__ table, uint32_t records, uint32_t per_lane, uint32_t *__restrict__ sink)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per_lane; k++) {
        const uint32_t r = mix(id * 0x9E3779B9u + k * 0x7F4A7C15u + 1u) % records;
        const uint4 v = table[r];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void k_gather48(const float4 *__restrict__ table, uint32_t records, uint32_t per_lane, uint32_t *__restrict__ sink)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (uint32_t k = 0; k < per_lane; k++) {
        const uint32_t r = mix(id * 0x9E3779B9u + k * 0x7F4A7C15u + 1u) % records;
        const float4 *p = (const float4 *)((const char *)table + (size_t)r * 48u);
        const float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 1.2345f) sink[0] = 1u;
}

int main()
{
    const size_t bytes = (size_t)6 << 30;            // 6 GiB table: far beyond L2 (32 MiB) and the 256 MiB die-level cache
    void *table; uint32_t *sink;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipMemset(sink, 0, 4));
    const uint32_t lanes = 256u * 4096u, per_lane = 64;
    const uint32_t rec16 = (uint32_t)(bytes / 16), rec48 = (uint32_t)(bytes / 48) - 1;
    hipLaunchKernelGGL(k_gather16, dim3(lanes / 256), dim3(256), 0, 0, (const uint4 *)table, rec16, per_lane, sink);
    hipLaunchKernelGGL(k_gather48, dim3(lanes / 256), dim3(256), 0, 0, (const float4 *)table, rec48, per_lane, sink);
    CHECK(hipDeviceSynchronize());
    const double loads = (double)lanes * per_lane;
    printf("k_gather16: %.0f gathers of 16 B (one 64-B line each: %.3f GB)\n", loads, loads * 64 / 1e9);
    printf("k_gather48: %.0f gathers of 48 B (1.5 64-B lines each on average: %.3f GB; 2 lines each: %.3f GB)\n", loads,
           loads * 96 / 1e9, loads * 128 / 1e9);
    return 0;
}
