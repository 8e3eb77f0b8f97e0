// Device memory of libvampnet_hip.so: EVERY allocation of the library goes through vn_dev_malloc / vn_dev_free (no hipMalloc anywhere
// else in csrc/), so that the whole path can run under GUARD PAGES (include/vampnet_hip_debug.h: vn_guard_mode, VN_GUARD_ALLOC).
//
// Guard mode (round 6; the memory-safety harness of tests/test_gpu_guard.py): a buffer is its own virtual-memory reservation
// (hipMemAddressReserve) with physical pages mapped (hipMemCreate / hipMemMap / hipMemSetAccess) over exactly the granules it needs and
// one UNMAPPED granule on each side:
//   mode 1 "end"   : the buffer's last byte (rounded up to 16 B: every kernel here makes 16-byte accesses) abuts the unmapped granule
//                    behind it — an over-read of one element faults instead of landing in a neighbour's pages;
//   mode 2 "start" : the buffer's first byte is the first byte of its first page — an under-read faults.
// The whole mapping — the buffer and the slack that rounds it to the granule — is POISON-filled (VN_GUARD_FILL, default 0xff: NaN as
// bf16 / fp16 / fp32, -1 as an integer): a kernel whose result depends on bytes nobody wrote (an uninitialised workspace row, the
// tail of a page behind a buffer) gives wrong numbers here instead of working by the accident of fresh pages being zero.
// A fault is the HSA runtime's "Memory access fault by GPU" abort of the process, which is what the harness asserts does NOT happen.
// hipMalloc, by contrast, sub-allocates from large mapped blocks: a kernel that reads a few KiB past a buffer is almost always silent
// (and once in a long while, when the buffer ends a block, it is an abort nobody can reproduce: profiles/r05_pytest_gpu_one_aborted_run.txt).
// The same two functions are exported in torch's pluggable-allocator signature (vn_guard_torch_alloc / vn_guard_torch_free), so a test
// process can put every torch tensor — inputs, outputs, weight blobs, planes, arenas — under the same regime.
#include <map>
#include <mutex>
#include <stdlib.h>
#include "vn_common.h"

namespace {
struct guard_block {
    void* base;                  // start of the reservation
    size_t reserved;             // bytes reserved (mapped + two guard granules)
    size_t mapped;
    hipMemGenericAllocationHandle_t handle;
    size_t bytes;                // what the caller asked for
};
std::mutex g_mu;
std::map<void*, guard_block> g_blocks;     // user pointer -> block
int g_mode = -1;                           // -1: not initialised (environment), 0 off, 1 end, 2 start
int g_fill = -1;                           // poison byte of fresh guard blocks (VN_GUARD_FILL; -1: not read yet; 256: leave as mapped)
int64_t g_count = 0, g_live = 0, g_live_bytes = 0;

int mode_now() {
    if (g_mode < 0) {
        const char* e = getenv("VN_GUARD_ALLOC");
        int m = 0;
        if (e && *e) m = (!strcmp(e, "end") || !strcmp(e, "1")) ? 1 : (!strcmp(e, "start") || !strcmp(e, "2")) ? 2 : 0;
        g_mode = m;
    }
    return g_mode;
}

// bisecting aids of the harness itself (scripts/guard_vmm_probe.py): VN_GUARD_PLAIN=1 backs a "guard" block with plain hipMalloc,
// VN_GUARD_NOFREE=1 never unmaps (no virtual or physical address is ever reused), VN_GUARD_SYNC=1 waits for the device after the fill
bool env_on(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }

hipError_t guard_alloc(void** out, size_t bytes, int mode) {
    *out = nullptr;
    static const bool plain = env_on("VN_GUARD_PLAIN"), sync_fill = env_on("VN_GUARD_SYNC");
    if (plain) {
        const hipError_t pe = hipMalloc(out, bytes ? bytes : 1);
        if (pe == hipSuccess) { std::lock_guard<std::mutex> g(g_mu); ++g_count; }
        return pe;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = 4096;
    const size_t want = bytes ? bytes : 1;
    const size_t need = (want + 15) & ~(size_t)15;                 // end mode: the 16-byte access that holds the last element is inside
    const size_t mapped = (need + gran - 1) / gran * gran;
    guard_block b = {};
    b.bytes = bytes; b.mapped = mapped; b.reserved = mapped + 2 * gran;
    e = hipMemAddressReserve(&b.base, b.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&b.handle, mapped, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(b.base, b.reserved); return e; }
    char* lo = (char*)b.base + gran;
    e = hipMemMap(lo, mapped, 0, b.handle, 0);
    if (e != hipSuccess) { (void)hipMemRelease(b.handle); (void)hipMemAddressFree(b.base, b.reserved); return e; }
    hipMemAccessDesc ad = {};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(lo, mapped, &ad, 1);
    if (e != hipSuccess) {
        (void)hipMemUnmap(lo, mapped); (void)hipMemRelease(b.handle); (void)hipMemAddressFree(b.base, b.reserved);
        return e;
    }
    if (g_fill < 0) {
        const char* f = getenv("VN_GUARD_FILL");
        g_fill = f && *f ? (!strcmp(f, "none") ? 256 : (int)(strtol(f, nullptr, 0) & 0xff)) : 0xff;
    }
    // The poison must be IN PLACE when the block is handed out.  hipMemset is asynchronous to the host and runs on the null stream: a
    // copy through the DMA engines or a kernel on a non-blocking stream that writes the fresh block can overtake it — the poison then
    // lands on top of real data (seen once in ~700 guarded tests: a mask tensor full of 0xff, a token -1 fetching row -1 of a table).  So
    // the fill runs on a stream of its own and the allocator waits for THAT stream only (not for the device: other work stays in flight).
    if (g_fill < 256) {
        static hipStream_t fill_stream = nullptr;
        if (!fill_stream && hipStreamCreateWithFlags(&fill_stream, hipStreamNonBlocking) != hipSuccess) fill_stream = nullptr;
        e = hipMemsetAsync(lo, g_fill, mapped, fill_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(fill_stream);
        if (e != hipSuccess) {
            (void)hipMemUnmap(lo, mapped); (void)hipMemRelease(b.handle); (void)hipMemAddressFree(b.base, b.reserved);
            return e;
        }
    }
    if (sync_fill) (void)hipDeviceSynchronize();
    void* user = mode == 2 ? (void*)lo : (void*)(lo + (mapped - need));
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_blocks[user] = b;
        ++g_count; ++g_live; g_live_bytes += (int64_t)mapped;
    }
    *out = user;
    return hipSuccess;
}

// returns false when p is not a guard block (a plain hipMalloc pointer: the mode was switched on later)
bool guard_free(void* p) {
    guard_block b;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_blocks.find(p);
        if (it == g_blocks.end()) return false;
        b = it->second;
        g_blocks.erase(it);
        --g_live; g_live_bytes -= (int64_t)b.mapped;
    }
    (void)hipDeviceSynchronize();                 // hipFree's contract: no kernel that may touch the block is in flight afterwards
    static const bool nofree = env_on("VN_GUARD_NOFREE");
    if (nofree) return true;
    char* lo = (char*)b.base + (b.reserved - b.mapped) / 2;
    // The pages go back, the ADDRESS RANGE does not (no hipMemAddressFree): (1) a freed block stays unmapped for the rest of the process,
    // so a use after free faults like an overrun does; (2) on this stack (ROCm 7.0.2 runtime inside torch 2.10, MI355X) a range that
    // is freed and reserved again reads STALE data through its new mapping — plain torch ops on such blocks return wrong numbers
    // (scripts/guard_vmm_probe.py: 10-17 mismatches per run with hipMemAddressFree, 0 without; profiles/r06_guard_vmm_probe.txt).
    // Virtual address space is the one thing a test process has plenty of.  VN_GUARD_REUSE_VA=1 restores the freeing (the probe's arm).
    static const bool reuse_va = env_on("VN_GUARD_REUSE_VA");
    (void)hipMemUnmap(lo, b.mapped);
    (void)hipMemRelease(b.handle);
    if (reuse_va) (void)hipMemAddressFree(b.base, b.reserved);
    return true;
}
}  // namespace

hipError_t vn_dev_malloc(void** p, size_t bytes) {
    static const bool lib_on = [] { const char* e = getenv("VN_GUARD_LIB"); return !(e && e[0] == '0'); }();   // bisecting aid
    const int m = lib_on ? mode_now() : 0;
    if (m == 0) return hipMalloc(p, bytes ? bytes : 1);
    return guard_alloc(p, bytes, m);
}

void vn_dev_free(void* p) {
    if (!p) return;
    if (guard_free(p)) return;
    (void)hipFree(p);
}

extern "C" int vn_guard_mode(int mode) {
    if (mode < 0 || mode > 2) return VN_ERR_INVALID;
    g_mode = mode;
    return VN_OK;
}

extern "C" int vn_guard_stats(int64_t* allocations, int64_t* live, int64_t* live_bytes) {
    std::lock_guard<std::mutex> g(g_mu);
    if (allocations) *allocations = g_count;
    if (live) *live = g_live;
    if (live_bytes) *live_bytes = g_live_bytes;
    return mode_now();
}

extern "C" int vn_guard_alloc(int64_t bytes, int mode, void** out) {
    if (bytes < 0 || !out || (mode != 1 && mode != 2)) return VN_ERR_INVALID;
    return guard_alloc(out, (size_t)bytes, mode) == hipSuccess ? VN_OK : VN_ERR_OOM;
}

extern "C" int vn_guard_free(void* p) { return guard_free(p) ? VN_OK : VN_ERR_INVALID; }

// torch.cuda.memory.CUDAPluggableAllocator(libvampnet_hip.so, "vn_guard_torch_alloc", "vn_guard_torch_free"): every torch allocation of
// the process becomes a guard block of the mode VN_GUARD_ALLOC names (default: end)
extern "C" void* vn_guard_torch_alloc(long size, int device, void* stream) {
    (void)stream;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    void* p = nullptr;
    const int m = mode_now();
    const hipError_t e = guard_alloc(&p, (size_t)(size > 0 ? size : 0), m == 0 ? 1 : m);
    if (cur != device) (void)hipSetDevice(cur);
    return e == hipSuccess ? p : nullptr;
}

extern "C" void vn_guard_torch_free(void* p, long size, int device, void* stream) {
    (void)size; (void)device; (void)stream;
    if (p && !guard_free(p)) (void)hipFree(p);
}

// self-test of the harness: ONE wave reads (write == 0) or writes one 32-bit word `offset_bytes` from p.  Past a guard block's end this
// must end the process (tests/test_gpu_guard.py runs it in a child and expects exactly that)
__global__ void vn_guard_poke_kernel(unsigned* p, long offset_words, int write, unsigned* sink) {
    if (threadIdx.x == 0) {
        if (write) p[offset_words] = 0xdeadbeefu;
        else *sink = p[offset_words];
    }
}
extern "C" int vn_guard_poke(void* p, int64_t offset_bytes, int write, void* sink, void* stream) {
    if (!p || !sink || (offset_bytes & 3)) return VN_ERR_INVALID;
    hipLaunchKernelGGL(vn_guard_poke_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned*)p, (long)(offset_bytes / 4), write, (unsigned*)sink);
    if (hipGetLastError() != hipSuccess) return VN_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? VN_OK : VN_ERR_HIP;
}
