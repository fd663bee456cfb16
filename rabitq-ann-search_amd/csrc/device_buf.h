// device_buf.h — RAII device buffer and the HIP error check shared by the C-ABI and the builder.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <stdexcept>
#include <string>

namespace cph {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            if (e_ == hipErrorOutOfMemory) throw std::bad_alloc();                        \
            throw ::cph::HipError(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + #expr); \
        }                                                                                 \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    void alloc(size_t count) {
        if (count <= n && p) return;
        release();
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
        n = count;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace cph
