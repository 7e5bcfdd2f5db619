/* TEST INFRASTRUCTURE -- never part of the product.
 *
 * A minimal HSA launcher for code objects built from the reference's own
 * OpenCL C kernel sources (oracle/build_ref_kernels.py compiles them, unmodified,
 * with ROCm's clang and its OpenCL built-in library into oracle/_ref/).  The
 * OpenCL runtime of this platform reports "Image support: No" for gfx950, so
 * the reference's host code (src/parallel.c, which creates 1-D image buffers,
 * src/parallel.c:282-324) cannot run here; its kernels can: a 1-D image-buffer
 * read compiles to a typed buffer load through a 4-dword buffer descriptor that
 * the kernel fetches from the address passed as the image argument.  This file
 * provides what a launcher needs -- device memory, copies, code-object loading
 * and one AQL kernel dispatch with a caller-packed kernarg block -- and nothing
 * of the path tracer.  Argument packing lives in oracle/refcl.py.
 *
 * Every entry point returns 0 on success and a negative number on failure;
 * hr_last_error() gives the text.
 */
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define HR_API __attribute__((visibility("default")))

static char g_err[512];
static int g_ready;
static hsa_agent_t g_gpu, g_cpu;
static int g_have_gpu, g_have_cpu;
static hsa_amd_memory_pool_t g_dev_pool, g_arg_pool;
static int g_have_dev_pool, g_have_arg_pool;
static hsa_queue_t *g_queue;

static int fail(const char *what, hsa_status_t st)
{
    const char *s = NULL;
    hsa_status_string(st, &s);
    snprintf(g_err, sizeof g_err, "%s: %s (0x%x)", what, s ? s : "?", (unsigned)st);
    return -1;
}
#define CHK(call) do { hsa_status_t st_ = (call); if (st_ != HSA_STATUS_SUCCESS) return fail(#call, st_); } while (0)

HR_API const char *hr_last_error(void) { return g_err; }

static hsa_status_t on_agent(hsa_agent_t a, void *data)
{
    (void)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = 1; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = 1; }
    return HSA_STATUS_SUCCESS;
}

static hsa_status_t on_dev_pool(hsa_amd_memory_pool_t p, void *data)
{
    (void)data;
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool can_alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &can_alloc);
    if (can_alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev_pool) {
        g_dev_pool = p; g_have_dev_pool = 1;
    }
    return HSA_STATUS_SUCCESS;
}

static hsa_status_t on_host_pool(hsa_amd_memory_pool_t p, void *data)
{
    (void)data;
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool can_alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &can_alloc);
    if (can_alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_arg_pool) {
        g_arg_pool = p; g_have_arg_pool = 1;
    }
    return HSA_STATUS_SUCCESS;
}

HR_API int hr_init(void)
{
    if (g_ready) return 0;
    CHK(hsa_init());
    CHK(hsa_iterate_agents(on_agent, NULL));
    if (!g_have_gpu || !g_have_cpu) { snprintf(g_err, sizeof g_err, "no GPU agent"); return -2; }
    CHK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_dev_pool, NULL));
    CHK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_host_pool, NULL));
    if (!g_have_dev_pool || !g_have_arg_pool) { snprintf(g_err, sizeof g_err, "memory pools not found"); return -3; }
    /* private/group segment sizes: UINT32_MAX lets the runtime size scratch per dispatch */
    CHK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, NULL, NULL, UINT32_MAX, UINT32_MAX, &g_queue));
    g_ready = 1;
    return 0;
}

HR_API int hr_agent_name(char *out64)
{
    if (!g_ready) return -1;
    CHK(hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, out64));
    return 0;
}

HR_API int hr_alloc(size_t bytes, void **out)
{
    *out = NULL;
    if (!g_ready) { snprintf(g_err, sizeof g_err, "hr_init not called"); return -1; }
    if (bytes == 0) bytes = 4;
    CHK(hsa_amd_memory_pool_allocate(g_dev_pool, bytes, 0, out));
    return 0;
}

HR_API int hr_free(void *p)
{
    if (p) CHK(hsa_amd_memory_pool_free(p));
    return 0;
}

/* Copies go through a host-pool staging buffer that both agents may touch. */
static int staged_copy(void *dst, const void *src, size_t bytes, int to_device)
{
    if (!bytes) return 0;
    void *stage = NULL;
    CHK(hsa_amd_memory_pool_allocate(g_arg_pool, bytes, 0, &stage));
    hsa_status_t st = hsa_amd_agents_allow_access(1, &g_gpu, NULL, stage);
    if (st != HSA_STATUS_SUCCESS) { hsa_amd_memory_pool_free(stage); return fail("allow_access(stage)", st); }
    hsa_signal_t sig;
    st = hsa_signal_create(1, 0, NULL, &sig);
    if (st != HSA_STATUS_SUCCESS) { hsa_amd_memory_pool_free(stage); return fail("signal_create", st); }
    if (to_device) memcpy(stage, src, bytes);
    st = to_device ? hsa_amd_memory_async_copy(dst, g_gpu, stage, g_cpu, bytes, 0, NULL, sig)
                   : hsa_amd_memory_async_copy(stage, g_cpu, src, g_gpu, bytes, 0, NULL, sig);
    int rc = 0;
    if (st != HSA_STATUS_SUCCESS) rc = fail("memory_async_copy", st);
    else {
        while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) { }
        if (!to_device) memcpy(dst, stage, bytes);
    }
    hsa_signal_destroy(sig);
    hsa_amd_memory_pool_free(stage);
    return rc;
}

HR_API int hr_h2d(void *dst, const void *src, size_t bytes) { return g_ready ? staged_copy(dst, src, bytes, 1) : -1; }
HR_API int hr_d2h(void *dst, const void *src, size_t bytes) { return g_ready ? staged_copy(dst, src, bytes, 0) : -1; }

HR_API int hr_fill32(void *dst, uint32_t value, size_t count)
{
    if (count) CHK(hsa_amd_memory_fill(dst, value, count));
    return 0;
}

typedef struct {
    hsa_executable_t exe;
    hsa_code_object_reader_t reader;
    void *image;
} hr_module;

HR_API int hr_load(const void *image, size_t bytes, void **out)
{
    *out = NULL;
    if (!g_ready) { snprintf(g_err, sizeof g_err, "hr_init not called"); return -1; }
    hr_module *m = calloc(1, sizeof *m);
    m->image = malloc(bytes);
    memcpy(m->image, image, bytes);
    CHK(hsa_code_object_reader_create_from_memory(m->image, bytes, &m->reader));
    CHK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, NULL, &m->exe));
    CHK(hsa_executable_load_agent_code_object(m->exe, g_gpu, m->reader, NULL, NULL));
    CHK(hsa_executable_freeze(m->exe, NULL));
    *out = m;
    return 0;
}

typedef struct {
    uint64_t object;
    uint32_t kernarg_size, group_size, private_size, dynamic_stack;
} hr_kernel;

HR_API int hr_kernel_info(void *module, const char *symbol, hr_kernel *k)
{
    hr_module *m = module;
    hsa_executable_symbol_t sym;
    CHK(hsa_executable_get_symbol_by_name(m->exe, symbol, &g_gpu, &sym));
    CHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k->object));
    CHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k->kernarg_size));
    CHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k->group_size));
    CHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k->private_size));
    bool dyn = false;
    CHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_DYNAMIC_CALLSTACK, &dyn));
    k->dynamic_stack = dyn ? 1u : 0u;
    return 0;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* One dispatch: kernarg is the complete kernel-argument block (explicit and
 * hidden arguments, packed by the caller from the code object's metadata).
 * Blocks until the kernel has finished or timeout_ms has passed (-9: the
 * kernel is still running and the process should end). */
HR_API int hr_dispatch(const hr_kernel *k, const void *kernarg, size_t kernarg_bytes, const uint32_t grid[3],
                       const uint32_t wg[3], double timeout_ms, double *elapsed_ms)
{
    if (!g_ready) { snprintf(g_err, sizeof g_err, "hr_init not called"); return -1; }
    if (k->dynamic_stack) { snprintf(g_err, sizeof g_err, "kernel needs a dynamic call stack"); return -4; }
    if (kernarg_bytes < k->kernarg_size) { snprintf(g_err, sizeof g_err, "kernarg block too small"); return -5; }
    for (int i = 0; i < 3; i++)
        if (wg[i] == 0 || grid[i] == 0 || grid[i] % wg[i]) { snprintf(g_err, sizeof g_err, "bad launch geometry"); return -6; }
    if ((uint64_t)wg[0] * wg[1] * wg[2] > 256) { snprintf(g_err, sizeof g_err, "workgroup too large"); return -6; }

    void *args = NULL;
    CHK(hsa_amd_memory_pool_allocate(g_arg_pool, kernarg_bytes < 64 ? 64 : kernarg_bytes, 0, &args));
    CHK(hsa_amd_agents_allow_access(1, &g_gpu, NULL, args));
    memcpy(args, kernarg, kernarg_bytes);
    hsa_signal_t done;
    CHK(hsa_signal_create(1, 0, NULL, &done));

    const uint64_t idx = hsa_queue_add_write_index_relaxed(g_queue, 1);
    hsa_kernel_dispatch_packet_t *pkt = (hsa_kernel_dispatch_packet_t *)g_queue->base_address + (idx & (g_queue->size - 1));
    memset((char *)pkt + 4, 0, sizeof *pkt - 4);
    pkt->workgroup_size_x = (uint16_t)wg[0]; pkt->workgroup_size_y = (uint16_t)wg[1]; pkt->workgroup_size_z = (uint16_t)wg[2];
    pkt->grid_size_x = grid[0]; pkt->grid_size_y = grid[1]; pkt->grid_size_z = grid[2];
    pkt->private_segment_size = k->private_size;
    pkt->group_segment_size = k->group_size;
    pkt->kernel_object = k->object;
    pkt->kernarg_address = args;
    pkt->completion_signal = done;
    const uint32_t dims = grid[2] > 1 ? 3 : (grid[1] > 1 ? 2 : 1);
    const uint16_t setup = (uint16_t)(dims << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                            (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    const double t0 = now_ms();
    __atomic_store_n((uint32_t *)pkt, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
    hsa_signal_store_screlease(g_queue->doorbell_signal, (hsa_signal_value_t)idx);

    int rc = 0;
    for (;;) {
        if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000, HSA_WAIT_STATE_BLOCKED) < 1) break;
        if (now_ms() - t0 > timeout_ms) { snprintf(g_err, sizeof g_err, "kernel still running after %.0f ms", timeout_ms); rc = -9; break; }
    }
    if (elapsed_ms) *elapsed_ms = now_ms() - t0;
    if (rc == 0) {
        hsa_signal_destroy(done);
        hsa_amd_memory_pool_free(args);
    }
    return rc;
}
