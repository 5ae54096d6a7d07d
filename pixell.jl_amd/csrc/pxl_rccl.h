// pxl_rccl.h -- the few RCCL entry points the sharded reprojection step and the pxl_comm_* helpers need, resolved at
// first use.  libpixell_hip.so has no link-time dependency on RCCL.
//
// Two situations, kept apart on purpose (a communicator is only meaningful to the RCCL instance that created it):
//   * the host already has an RCCL instance in the process (torch ships its own librccl.so; a Julia or C host may
//     have linked one) and hands pxl_reproject_sharded_step_* a communicator of THAT instance: the functions are
//     looked up in the library that is already loaded (dlopen RTLD_NOLOAD), or in PXL_RCCL_LIB when the host names
//     it explicitly.  Nothing is ever loaded implicitly on behalf of a foreign communicator.
//   * the host has no RCCL of its own and creates its communicator through pxl_comm_init_rank: the library then
//     loads librccl.so itself ("own" instance) and remembers the communicators it created; the sharded step accepts
//     only those while running on an instance it loaded itself.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only
#include <mutex>
#include <set>

struct RcclApi {
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    const char* (*GetErrorString)(ncclResult_t);
    char where[256];
    bool ok;
    bool own;       // loaded by this library itself (no instance was present): foreign communicators are refused
};

static void rccl_resolve(RcclApi& a, void* h) {
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.ok = a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.CommCount && a.CommUserRank && a.GetUniqueId &&
           a.CommInitRank && a.CommDestroy && a.GetErrorString;
}

static std::mutex g_rccl_mu;
static RcclApi g_rccl = {};
static bool g_rccl_tried_own = false;
static std::set<void*> g_own_comms;

// allow_own: the caller is about to CREATE a communicator, so loading an instance of our own is legitimate.
// Returns a COPY taken under the lock (the table is frozen once ok; until then a later call may still fill it in).
// A negative probe for an instance already in the process is NOT remembered: it is one cheap dlopen(RTLD_NOLOAD),
// and a host may load its RCCL after an early diagnostic call (pxl_comm_backend) or a step issued too soon.
static RcclApi rccl_api(bool allow_own) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.ok) return g_rccl;
    {
        const char* env = getenv("PXL_RCCL_LIB");
        void* h = nullptr;
        if (env && *env) {                                   // explicit: the host says which instance is its own
            h = dlopen(env, RTLD_NOW);
            if (h) snprintf(g_rccl.where, sizeof g_rccl.where, "%s (PXL_RCCL_LIB)", env);
            else snprintf(g_rccl.where, sizeof g_rccl.where, "PXL_RCCL_LIB=%s: %s", env, dlerror());
        } else {
            for (const char* nm : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);      // only an instance the process already holds
                if (h) { snprintf(g_rccl.where, sizeof g_rccl.where, "%s (already loaded)", nm); break; }
            }
            if (!h) snprintf(g_rccl.where, sizeof g_rccl.where, "no RCCL instance is loaded in this process (set PXL_RCCL_LIB, or create the communicator with pxl_comm_init_rank)");
        }
        if (h) { rccl_resolve(g_rccl, h); g_rccl.own = false; if (!g_rccl.ok) snprintf(g_rccl.where, sizeof g_rccl.where, "RCCL instance lacks a required entry point"); }
    }
    if (!g_rccl.ok && allow_own && !g_rccl_tried_own) {
        g_rccl_tried_own = true;
        void* h = nullptr;
        for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(nm, RTLD_NOW);
            if (h) { snprintf(g_rccl.where, sizeof g_rccl.where, "%s (loaded by libpixell_hip)", nm); break; }
        }
        if (h) { rccl_resolve(g_rccl, h); g_rccl.own = true; }
        else snprintf(g_rccl.where, sizeof g_rccl.where, "librccl.so not found (%s)", dlerror());
    }
    return g_rccl;
}

static void rccl_own_comm_add(void* c) { std::lock_guard<std::mutex> lock(g_rccl_mu); g_own_comms.insert(c); }
static bool rccl_own_comm_erase(void* c) { std::lock_guard<std::mutex> lock(g_rccl_mu); return g_own_comms.erase(c) > 0; }
static bool rccl_own_comm_has(void* c) { std::lock_guard<std::mutex> lock(g_rccl_mu); return g_own_comms.count(c) > 0; }
