// Thin C++ face of the C ABI (include/ptgpu.h) for the host classes: one lazily created context per process/device,
// error codes turned into exceptions of type FW::GpuError.  There is no CPU fallback behind these calls.
#pragma once
#include <stdexcept>
#include <string>

#include "../../../include/ptgpu.h"

namespace FW {

struct GpuError : std::runtime_error {
    explicit GpuError(const std::string& what) : std::runtime_error(what) {}
};

inline void gpuCheck(int rc, const char* call) {
    if (rc != PTGPU_OK) throw GpuError(std::string(call) + ": " + ptgpu_last_error());
}
#define PT_GPU(call) ::FW::gpuCheck((call), #call)

class GpuDevice {
public:
    // device index: PTGPU_DEVICE env var, else LOCAL_RANK (one process per GPU), else 0
    static GpuDevice& instance();
    ptgpu_context* ctx() { return m_ctx; }
    ~GpuDevice() { if (m_ctx) ptgpu_shutdown(m_ctx); }

private:
    GpuDevice();
    ptgpu_context* m_ctx = nullptr;
};

// RAII device buffer
template <class T> class GpuBuffer {
public:
    GpuBuffer() {}
    explicit GpuBuffer(size_t n) { resize(n); }
    ~GpuBuffer() { release(); }
    GpuBuffer(const GpuBuffer&) = delete;
    GpuBuffer& operator=(const GpuBuffer&) = delete;
    void resize(size_t n) {
        if (n <= m_cap) { m_n = n; return; }
        release();
        void* p = nullptr;
        PT_GPU(ptgpu_malloc(GpuDevice::instance().ctx(), n * sizeof(T), &p));
        m_p = (T*)p; m_n = m_cap = n;
    }
    void release() { if (m_p) ptgpu_free(GpuDevice::instance().ctx(), m_p); m_p = nullptr; m_n = m_cap = 0; }
    void upload(const T* h, size_t n) { resize(n); PT_GPU(ptgpu_memcpy_h2d(GpuDevice::instance().ctx(), m_p, h, n * sizeof(T))); }
    void download(T* h, size_t n) const { PT_GPU(ptgpu_memcpy_d2h(GpuDevice::instance().ctx(), h, m_p, n * sizeof(T))); }
    void zero() { if (m_n) PT_GPU(ptgpu_memset(GpuDevice::instance().ctx(), m_p, 0, m_n * sizeof(T))); }
    T* ptr() const { return m_p; }
    size_t size() const { return m_n; }

private:
    T* m_p = nullptr;
    size_t m_n = 0, m_cap = 0;
};

}  // namespace FW
