// common.hpp -- error plumbing and small RAII helpers shared by every libmfx translation unit.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mfx.h"

namespace mfx {

// Thread-local message behind mfx_last_error().
std::string& last_error();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Evaluates a HIP call; on failure records "<what>: <hip error>" and returns MFX_ERR_HIP from
// the enclosing function (the reference printed "GPUassert: ..." and carried on,
// cuda_src/CUDA_AUX.h:11-18; here the error is returned to the caller instead).
#define MFX_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::mfx::fail(MFX_ERR_HIP, "%s failed: %s (%s:%d)", #call,                  \
                               hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

#define MFX_TRY(call)                \
    do {                             \
        int _s = (call);             \
        if (_s != MFX_OK) return _s; \
    } while (0)

#define MFX_REQUIRE(cond, ...)                                      \
    do {                                                            \
        if (!(cond)) return ::mfx::fail(MFX_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// Selects `device` after checking that it exists; MFX_ERR_NO_DEVICE otherwise.  There is no
// CPU fallback anywhere in the library: every compute entry point starts with this call.
int use_device(int device);

// Owning device allocation.  Not copyable; release() is idempotent.
template <typename T>
class DevBuf {
public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DevBuf() { release(); }

    int alloc(size_t n) {
        release();
        if (n == 0) return MFX_OK;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T));
        if (e != hipSuccess) {
            p_ = nullptr;
            return fail(MFX_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        }
        n_ = n;
        return MFX_OK;
    }
    int alloc_zero(size_t n, hipStream_t st) {
        MFX_TRY(alloc(n));
        if (n) MFX_HIP(hipMemsetAsync(p_, 0, n * sizeof(T), st));
        return MFX_OK;
    }
    // Copies n elements from `src` living in `space`.
    int upload(const T* src, size_t n, mfx_memspace space, hipStream_t st) {
        if (n == 0) return MFX_OK;
        MFX_HIP(hipMemcpyAsync(p_, src, n * sizeof(T),
                               space == MFX_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
        return MFX_OK;
    }
    void release() {
        if (p_) (void) hipFree(p_);
        p_ = nullptr;
        n_ = 0;
    }
    T* get() const { return p_; }
    size_t size() const { return n_; }

private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t) ((a + b - 1) / b); }

}  // namespace mfx
