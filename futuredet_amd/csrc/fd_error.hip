// Process-level pieces of the C ABI: thread-local last-error text (never exit()), per-device caches and the
// tuning knobs.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "fd_common.h"

namespace fd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return dev;
}

int device_cu_count() {
    static std::atomic<int> cache[kMaxDevices];  // zero-initialised; racing first calls store the same value
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

bool ensure_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done) {
    const int dev = current_device();
    const uint64_t bit = (dev >= 0 && dev < kMaxDevices) ? (1ull << dev) : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return true;
    // idempotent: two threads racing here both set the same attribute value
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return true;
}

namespace {
struct TuneTable {
    std::atomic<int> v[kTuneCount];
    TuneTable() {
        static const char *const env[kTuneCount] = {"FD_SPCONV_RG", "FD_SPCONV_V1", "FD_SPCONV_BF16_V1", "FD_V2_DEPTH", "FD_V2_TM",
                                                    "FD_V2_LDSPAD", "FD_CONV_NT", "FD_V2_RANGES_PER_CU", "FD_V2_UNIFORM", "FD_V2_ROWCOST", "FD_SPCONV_C32", "FD_BF16_GP", "FD_BF16_RG", "FD_BF16_DEPTH", "FD_BF16_NW", "FD_STRICT", "FD_BF16_WIN", "FD_F32_RES_RG", "FD_CONV_STRIP", "FD_F32_RES_NW"};
        for (int i = 0; i < kTuneCount; ++i) {
            const char *e = getenv(env[i]);
            v[i].store(e ? atoi(e) : 0, std::memory_order_relaxed);
        }
    }
};
TuneTable &tune_table() {
    static TuneTable t;  // constructed once, thread-safe (C++11); the environment is read here and nowhere else
    return t;
}
const char *const kTuneNames[kTuneCount] = {"spconv_rg", "spconv_v1", "spconv_bf16_v1", "v2_depth", "v2_tm", "v2_ldspad", "conv_nt", "v2_ranges_per_cu", "v2_uniform", "v2_rowcost", "spconv_c32", "bf16_gp", "bf16_rg", "bf16_depth", "bf16_nw", "strict", "bf16_win", "f32_res_rg", "conv_strip", "f32_res_nw"};
}  // namespace

int tuning(TuneKey key) { return tune_table().v[key].load(std::memory_order_relaxed); }
}  // namespace fd

namespace {
// up to three regions per launch, 16-byte stores (region starts are 16-byte aligned in every caller's layout; a tail or a
// misaligned head of < 4 words goes out as single words)
struct FillJob {
    uint32_t *p[3];
    uint32_t v[3];
    size_t n[3];
    unsigned first_block[4];  // region r owns blocks [first_block[r], first_block[r + 1])
};
__global__ void __launch_bounds__(256) fill_words_kernel(FillJob j) {
    const int r = blockIdx.x >= j.first_block[2] ? 2 : (blockIdx.x >= j.first_block[1] ? 1 : 0);
    uint32_t *__restrict__ p = j.p[r];
    const uint32_t v = j.v[r];
    const size_t n = j.n[r];
    const size_t nb = j.first_block[r + 1] - j.first_block[r], b = blockIdx.x - j.first_block[r];
    const size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4 < n ? ((16 - ((uintptr_t)p & 15)) & 15) / 4 : n;
    const size_t n4 = (n - head) / 4;
    uint4 *p4 = reinterpret_cast<uint4 *>(p + head);
    const uint4 v4 = make_uint4(v, v, v, v);
    for (size_t i = b * 256 + threadIdx.x; i < n4; i += nb * 256) p4[i] = v4;
    if (b == 0) {
        if (threadIdx.x < head) p[threadIdx.x] = v;
        const size_t tail0 = head + n4 * 4;
        if (tail0 + threadIdx.x < n && threadIdx.x < 4) p[tail0 + threadIdx.x] = v;
    }
}
}  // namespace

int fd::fill_words3(void *p0, uint32_t v0, size_t n0, void *p1, uint32_t v1, size_t n1, void *p2, uint32_t v2, size_t n2, hipStream_t stream) {
    FillJob j;
    void *ps[3] = {p0, p1, p2};
    const uint32_t vs[3] = {v0, v1, v2};
    const size_t ns[3] = {n0, n1, n2};
    unsigned total = 0;
    for (int r = 0; r < 3; ++r) {
        j.p[r] = (uint32_t *)ps[r];
        j.v[r] = vs[r];
        j.n[r] = ps[r] ? ns[r] : 0;
        j.first_block[r] = total;
        size_t blocks = j.n[r] ? (j.n[r] / 4 + 1023) / 1024 : 0;  // four 16-byte stores per thread
        if (j.n[r] && blocks == 0) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        total += (unsigned)blocks;
    }
    j.first_block[3] = total;
    if (!total) return 0;
    hipLaunchKernelGGL(fill_words_kernel, dim3(total), dim3(256), 0, stream, j);
    return 0;
}

int fd::fill_words(void *p, uint32_t value, size_t n_words, hipStream_t stream) {
    return fd::fill_words3(p, value, n_words, nullptr, 0u, 0, nullptr, 0u, 0, stream);
}

extern "C" const char *fd_last_error(void) { return fd::g_err; }
extern "C" int fd_abi_version(void) { return 8; }

extern "C" int fd_tuning_set(const char *name, int value) {
    FD_REQUIRE(name, "fd_tuning_set: null name");
    for (int i = 0; i < fd::kTuneCount; ++i)
        if (strcmp(name, fd::kTuneNames[i]) == 0) {
            fd::tune_table().v[i].store(value, std::memory_order_relaxed);
            return FD_OK;
        }
    fd::set_error("fd_tuning_set: unknown knob '%s'", name);
    return FD_EINVAL;
}
