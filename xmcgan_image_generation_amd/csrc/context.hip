// Per-device context handle of the C ABI (include/xmcgan_hip.h: xmc_create / xmc_destroy).
//
// The launch entry points keep no mutable global state except idempotent, lock-free per-device flags (the opt-in
// to > 64 KiB of dynamic LDS, common.h::XmcLdsOptIn), so a handle is not REQUIRED to launch; creating one
// validates the device (gfx950 only) and performs that per-device setup eagerly -- e.g. before a hipGraph capture
// or before several host threads start issuing work for the same GPU.
#include <cstring>
#include <new>

#include "common.h"

struct xmc_context {
    int device;
    char arch[64];
};

extern "C" int xmc_create(int32_t device, void** handle) {
    XMC_REQUIRE(handle);
    *handle = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return xmc_hip_err(e);
    XMC_REQUIRE(device >= 0 && device < count);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return xmc_hip_err(e);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return XMC_EINVAL;          // CDNA4 only
    int prev = 0;
    e = hipGetDevice(&prev);
    if (e != hipSuccess) return xmc_hip_err(e);
    e = hipSetDevice(device);
    if (e != hipSuccess) return xmc_hip_err(e);
    int rc = xmc_internal_optin_conv_stream();
    if (rc == XMC_OK) rc = xmc_internal_optin_wgrad_dma();
    if (rc == XMC_OK) rc = xmc_internal_optin_wgrad_patch();
    if (rc == XMC_OK) rc = xmc_internal_optin_losses();
    (void)hipSetDevice(prev);
    if (rc != XMC_OK) return rc;
    xmc_context* c = new (std::nothrow) xmc_context;
    if (!c) return -12;                                                                 // -ENOMEM
    c->device = device;
    std::strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
    c->arch[sizeof(c->arch) - 1] = 0;
    *handle = c;
    return XMC_OK;
}

extern "C" int xmc_destroy(void* handle) {
    XMC_REQUIRE(handle);
    delete static_cast<xmc_context*>(handle);
    return XMC_OK;
}

extern "C" int xmc_handle_device(void* handle) {
    XMC_REQUIRE(handle);
    return static_cast<xmc_context*>(handle)->device;
}
