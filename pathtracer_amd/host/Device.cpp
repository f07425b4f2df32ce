#include <cstdlib>

#include "gpu/Device.hpp"

namespace FW {

GpuDevice::GpuDevice() {
    int dev = 0;
    if (const char* e = std::getenv("PTGPU_DEVICE")) dev = std::atoi(e);
    else if (const char* r = std::getenv("LOCAL_RANK")) dev = std::atoi(r);
    PT_GPU(ptgpu_init(dev, &m_ctx));
}

GpuDevice& GpuDevice::instance() {
    static GpuDevice d;
    return d;
}

}  // namespace FW
