// SyncBatchNorm peer mailbox: the memory behind tcvom_bn_finalize_sync / tcvom_bn_bwd_finalize_sync (norm.hip: bn_sync_exchange).
// Replaces the ~370 small NCCL collectives per step that nn.SyncBatchNorm issues under train_ddp.py:271-280.
//
// The ONLY entry points of the library that own memory: the mailbox must be UNCACHED device memory (remote stores over xGMI land
// in HBM, local polls must not be served from a stale L2 line -- what the collective libraries use for their flag buffers), and
// it must be exportable through hipIpc, neither of which the caller's PyTorch allocator provides.  One process per GPU:
//   rank r:  tcvom_mbox_alloc -> 64-byte handle -> exchanged by the host (torch.distributed all_gather_object)
//            tcvom_mbox_open(handle of rank q) for every q != r  -> device table of `world` base pointers (tcvom_bn_sync.peers)
#include "common.h"
#include <string.h>

extern "C" int tcvom_mbox_alloc(int64_t bytes, void** ptr, void* handle64) {
    TCVOM_CHECK_ARG(bytes > 0 && ptr && handle64, "mbox_alloc: bad args");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) return tcvom_fail(TCVOM_ERR_LAUNCH, "mbox_alloc: hipExtMallocWithFlags(%lld bytes): %s", (long long)bytes, hipGetErrorString(e));
    e = hipMemset(p, 0, (size_t)bytes);                       // tag 0 = "never written"
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return tcvom_fail(TCVOM_ERR_LAUNCH, "mbox_alloc: memset: %s", hipGetErrorString(e)); }
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        // a one-rank (loop-back) mailbox still works without an exportable handle; the caller sees an all-zero handle
        (void)hipGetLastError();
        memset(&h, 0, sizeof(h));
    }
    memcpy(handle64, &h, 64);
    *ptr = p;
    return TCVOM_OK;
}

extern "C" int tcvom_mbox_open(const void* handle64, void** ptr) {
    TCVOM_CHECK_ARG(handle64 && ptr, "mbox_open: bad args");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "mbox_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
    *ptr = p;
    return TCVOM_OK;
}

extern "C" int tcvom_mbox_close(void* ptr) {
    if (!ptr) return TCVOM_OK;
    const hipError_t e = hipIpcCloseMemHandle(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "mbox_close: %s", hipGetErrorString(e)); }
    return TCVOM_OK;
}

extern "C" int tcvom_mbox_free(void* ptr) {
    if (!ptr) return TCVOM_OK;
    const hipError_t e = hipFree(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "mbox_free: %s", hipGetErrorString(e)); }
    return TCVOM_OK;
}

extern "C" int tcvom_mbox_device(const void* ptr, int32_t* device) {
    TCVOM_CHECK_ARG(ptr && device, "mbox_device: bad args");
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return tcvom_fail(TCVOM_ERR_LAUNCH, "mbox_device: hipPointerGetAttributes: %s", hipGetErrorString(e)); }
    *device = (int32_t)at.device;
    return TCVOM_OK;
}
