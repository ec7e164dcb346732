// RCCL communicator behind the C ABI (xmca_comm_*): the ONE collective of the path - the all-gather of the per-rank spectra of
// a run-sharded rule_n (SURVEY 8(b) `mca_comm_*`, 8(e); the reference is single-process, xmca/array.py:1753-1771 is the loop that
// is sharded).  One process per GPU; rank 0 makes a unique id (xmca_comm_unique_id), the caller ships its 128 bytes to the
// other ranks by whatever it has (a file, MPI, a torch store), every rank calls xmca_comm_create.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME torch's own copy carries too - a process that has already
// loaded one gets that one back, so there is never a second RCCL in the process) and only its types come from <rccl/rccl.h>:
// the library loads and every other entry point works on a box without RCCL.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace xmca {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string why;

  static RcclApi& get() {
    static RcclApi* api = [] {
      RcclApi* a = new RcclApi;
      const char* names[] = {std::getenv("XMCA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names) {
        if (!n || !n[0]) continue;
        a->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a->lib) break;
        a->why = dlerror();
      }
      if (!a->lib) return a;
      bool ok = true;
      auto sym = [&](const char* name) {
        void* p = dlsym(a->lib, name);
        if (!p) { ok = false; a->why = std::string("missing symbol ") + name; }
        return p;
      };
      a->GetUniqueId = reinterpret_cast<decltype(a->GetUniqueId)>(sym("ncclGetUniqueId"));
      a->CommInitRank = reinterpret_cast<decltype(a->CommInitRank)>(sym("ncclCommInitRank"));
      a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(sym("ncclCommDestroy"));
      a->AllGather = reinterpret_cast<decltype(a->AllGather)>(sym("ncclAllGather"));
      a->Broadcast = reinterpret_cast<decltype(a->Broadcast)>(sym("ncclBroadcast"));
      a->GetErrorString = reinterpret_cast<decltype(a->GetErrorString)>(sym("ncclGetErrorString"));
      a->GetVersion = reinterpret_cast<decltype(a->GetVersion)>(sym("ncclGetVersion"));
      if (!ok) { dlclose(a->lib); a->lib = nullptr; }
      return a;
    }();
    return *api;
  }
  void require() const {
    XMCA_CHECK(lib != nullptr, XMCA_ERR_UNSUPPORTED, "RCCL is not available in this process (librccl.so.1: " + why + ")");
  }
  void check(ncclResult_t r, const char* what) const {
    if (r != ncclSuccess)
      throw Error(XMCA_ERR_HIP, std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
  }
};

static_assert(XMCA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "xmca_comm_unique_id hands out an ncclUniqueId");

}  // namespace xmca

struct xmca_comm {
  int device = 0;
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;
  ::xmca::DevBuf<double> send, recv;
  std::string err;
  long long collectives = 0;       // all-gathers / broadcasts carried out
  long long bytes = 0;             // bytes this rank received in them
};
