// pxl_rccl.h -- the few RCCL entry points the sharded reprojection step needs, resolved at first use.
// libpixell_hip.so has no link-time dependency on RCCL: a host that shards across GPUs already has an RCCL
// instance in its process (torch ships its own librccl.so; a Julia or C host links one), and the communicator
// handed to pxl_reproject_sharded_step_* belongs to THAT instance, so its functions are looked up in the library
// that is already loaded (RTLD_NOLOAD) before anything else is tried.  PXL_RCCL_LIB overrides the name.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only

struct RcclApi {
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
    const char* (*GetErrorString)(ncclResult_t);
    char where[256];
    bool ok;
};

static RcclApi rccl_load() {
    RcclApi a = {};
    const char* env = getenv("PXL_RCCL_LIB");
    const char* names[] = {env && *env ? env : "librccl.so", "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)          // first: whatever instance the process already holds
        for (const char* nm : names) {
            h = dlopen(nm, RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { snprintf(a.where, sizeof a.where, "%s%s", nm, pass == 0 ? " (already loaded)" : ""); break; }
        }
    if (!h) { snprintf(a.where, sizeof a.where, "librccl.so not found (%s)", dlerror()); return a; }
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.ok = a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.CommCount && a.CommUserRank && a.GetErrorString;
    return a;
}

static const RcclApi& rccl_api() {
    static const RcclApi api = rccl_load();
    return api;
}
