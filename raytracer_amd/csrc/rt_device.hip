// Device, stream, memory and event plumbing behind rt_hip.h.
#include "rt_internal.hpp"

#include <cstring>

static const char *k_kernel_names[RT_K_COUNT] = {
    "generate_rays", "kd_intersect", "spath_init", "spath_trace", "xorshift",
    "resolve_avg", "cast_rays", "path_trace", "f_buffer_average", "f_to_byte", "spath_run", "pixel_order", "debug_draw"};

extern "C" {

int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

rt_device *rt_device_open(int ordinal)
{
    int n = rt_device_count();
    if (n <= 0) {
        fprintf(stderr, "ERROR: no HIP device is visible; this library has no CPU fallback\n");
        exit(1);
    }
    if (ordinal < 0 || ordinal >= n) {
        fprintf(stderr, "ERROR: device ordinal %d out of range (%d visible)\n", ordinal, n);
        exit(1);
    }
    rt_device *dev = new rt_device();
    dev->ordinal = ordinal;
    RT_HIP_CHECK(hipSetDevice(ordinal));
    RT_HIP_CHECK(hipGetDeviceProperties(&dev->props, ordinal));
    dev->num_streams = RT_MAX_STREAMS;
    for (int i = 0; i < dev->num_streams; i++)
        RT_HIP_CHECK(hipStreamCreateWithFlags(&dev->streams[i], hipStreamNonBlocking));
    dev->current = 0;
    dev->stream = dev->streams[0];
    RT_HIP_CHECK(hipMalloc((void **)&dev->counters, 64 * RT_MAX_STREAMS));
    RT_HIP_CHECK(hipMemset(dev->counters, 0, 64 * RT_MAX_STREAMS));
    rt_kernels_setup_device(dev);
    return dev;
}

void rt_device_close(rt_device *dev)
{
    if (!dev) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    rt_device_sync(dev);
    for (int k = 0; k < RT_K_COUNT; k++) {
        for (hipEvent_t e : dev->slots[k].start) (void)hipEventDestroy(e);
        for (hipEvent_t e : dev->slots[k].stop) (void)hipEventDestroy(e);
    }
    for (hipEvent_t e : dev->free_events) (void)hipEventDestroy(e);
    for (int i = 0; i < dev->num_streams; i++) RT_HIP_CHECK(hipStreamDestroy(dev->streams[i]));
    RT_HIP_CHECK(hipFree(dev->counters));
    rt_kernels_release_device(dev);
    delete dev;
}

void rt_device_get_props(rt_device *dev, rt_device_props *out)
{
    memset(out, 0, sizeof *out);
    // Some driver stacks leave the marketing name empty; the ISA name is always there.
    const char *name = dev->props.name[0] ? dev->props.name : dev->props.gcnArchName;
    strncpy(out->name, name, sizeof(out->name) - 1);
    out->compute_units = dev->props.multiProcessorCount;
    out->wave_size = dev->props.warpSize;
    out->max_waves_per_cu = dev->props.maxThreadsPerMultiProcessor / (dev->props.warpSize ? dev->props.warpSize : 64);
    out->global_mem_bytes = dev->props.totalGlobalMem;
    out->lds_bytes_per_cu = dev->props.maxSharedMemoryPerMultiProcessor;
}

void rt_device_sync(rt_device *dev)
{
    for (int i = 0; i < dev->num_streams; i++) RT_HIP_CHECK(hipStreamSynchronize(dev->streams[i]));
}

void rt_device_select_stream(rt_device *dev, int index)
{
    if (index < 0 || index >= dev->num_streams) {
        fprintf(stderr, "ERROR: stream index %d out of range (%d streams)\n", index, dev->num_streams);
        exit(1);
    }
    dev->current = index;
    dev->stream = dev->streams[index];
}

void *rt_device_stream(rt_device *dev) { return (void *)dev->stream; }
int rt_device_ordinal(rt_device *dev) { return dev->ordinal; }

rt_mem rt_mem_alloc(rt_device *dev, size_t bytes)
{
    void *p = nullptr;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    RT_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    return p;
}

void rt_mem_free(rt_device *dev, rt_mem m)
{
    if (!m) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    rt_device_sync(dev);
    RT_HIP_CHECK(hipFree(m));
}

void rt_mem_upload(rt_device *dev, rt_mem dst, const void *src, size_t bytes)
{
    if (!bytes) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    rt_device_sync(dev);       /* blocking transfer: ordered behind everything queued on any stream */
    RT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, dev->stream));
    RT_HIP_CHECK(hipStreamSynchronize(dev->stream));
}

void rt_mem_download(rt_device *dev, void *dst, rt_mem src, size_t bytes)
{
    if (!bytes) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    rt_device_sync(dev);
    RT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, dev->stream));
    RT_HIP_CHECK(hipStreamSynchronize(dev->stream));
}

void *rt_host_alloc(rt_device *dev, size_t bytes)
{
    void *p = nullptr;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    RT_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault));
    return p;
}

void rt_host_free(rt_device *dev, void *p)
{
    if (!p) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    rt_device_sync(dev);
    RT_HIP_CHECK(hipHostFree(p));
}

void rt_mem_upload_queued(rt_device *dev, rt_mem dst, const void *pinned_src, size_t bytes)
{
    if (!bytes) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    RT_HIP_CHECK(hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, dev->stream));
}

void rt_event_wait(rt_device *dev, rt_event e)
{
    (void)dev;
    if (e) RT_HIP_CHECK(hipEventSynchronize((hipEvent_t)e));
}

void rt_mem_zero(rt_device *dev, rt_mem m, size_t bytes)
{
    if (!bytes) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    RT_HIP_CHECK(hipMemsetAsync(m, 0, bytes, dev->stream));
}

void rt_mem_copy(rt_device *dev, rt_mem dst, rt_mem src, size_t bytes)
{
    if (!bytes) return;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    RT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dev->stream));
}

void rt_timing_enable(rt_device *dev, int on) { dev->timing = on ? 1 : 0; }

void rt_timing_reset(rt_device *dev)
{
    rt_device_sync(dev);
    for (int k = 0; k < RT_K_COUNT; k++) {
        rt_timing_fold(dev, k);
        dev->slots[k].launches = 0;
        dev->slots[k].total_ms = 0.0;
    }
}

void rt_timing_get(rt_device *dev, int kernel_id, uint64_t *launches, double *total_ms)
{
    if (kernel_id < 0 || kernel_id >= RT_K_COUNT) { *launches = 0; *total_ms = 0; return; }
    rt_device_sync(dev);
    rt_timing_fold(dev, kernel_id);
    *launches = dev->slots[kernel_id].launches;
    *total_ms = dev->slots[kernel_id].total_ms;
}

rt_event rt_event_record(rt_device *dev)
{
    hipEvent_t e;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));      /* events belong to the device that is current when they are created */
    RT_HIP_CHECK(hipEventCreate(&e));
    RT_HIP_CHECK(hipEventRecord(e, dev->stream));
    return (rt_event)e;
}

float rt_event_elapsed_ms(rt_device *dev, rt_event start, rt_event stop)
{
    (void)dev;
    float ms = 0.f;
    RT_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    RT_HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ms;
}

void rt_event_release(rt_device *dev, rt_event e)
{
    (void)dev;
    if (e) RT_HIP_CHECK(hipEventDestroy((hipEvent_t)e));
}

const char *rt_kernel_name(int kernel_id)
{
    return (kernel_id >= 0 && kernel_id < RT_K_COUNT) ? k_kernel_names[kernel_id] : "?";
}

}  // extern "C"

// Adds the finished event pairs of one kernel to its totals and recycles the
// events.  The stream must have been synchronised.
void rt_timing_fold(rt_device *dev, int kernel_id)
{
    rt_timing_slot &s = dev->slots[kernel_id];
    for (size_t i = 0; i < s.start.size(); i++) {
        float ms = 0.f;
        RT_HIP_CHECK(hipEventElapsedTime(&ms, s.start[i], s.stop[i]));
        s.total_ms += ms;
        s.launches++;
        dev->free_events.push_back(s.start[i]);
        dev->free_events.push_back(s.stop[i]);
    }
    s.start.clear();
    s.stop.clear();
}

static hipEvent_t take_event(rt_device *dev)
{
    if (!dev->free_events.empty()) {
        hipEvent_t e = dev->free_events.back();
        dev->free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    RT_HIP_CHECK(hipEventCreate(&e));
    return e;
}

rt_launch_timer::rt_launch_timer(rt_device *d, int kernel_id) : dev(d), id(kernel_id)
{
    // every launcher constructs one of these right before its launch: a process may hold rt_devices on several
    // GPUs, and a launch (or an event) goes to the device that is current
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    if (!dev->timing) return;
    if (dev->slots[id].start.size() >= 16384) {   // bound the number of live events
        rt_device_sync(dev);
        rt_timing_fold(dev, id);
    }
    e0 = take_event(dev);
    e1 = take_event(dev);
    RT_HIP_CHECK(hipEventRecord(e0, dev->stream));
}

rt_launch_timer::~rt_launch_timer()
{
    if (!e0) return;
    RT_HIP_CHECK(hipEventRecord(e1, dev->stream));
    dev->slots[id].start.push_back(e0);
    dev->slots[id].stop.push_back(e1);
}
