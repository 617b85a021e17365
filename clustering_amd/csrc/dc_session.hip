// dc_session.hip -- a trajectory kept RESIDENT on the GPUs of this process across the phases of the
// density path: coordinates are uploaded once per device, the operand
// workspace, the populations, the free energies and the neighbour arrays stay in HBM from
//     pop -> FE -> NN -> sigma2 -> (second pop + NN at the lumping radius) -> radius forest,
// which is the flow of density_clustering.cpp:597-817.  The reference's multi-GPU host code
// (density_clustering_cuda.cu:139-182, :286-328) runs one OpenMP thread per device, copies every
// partial result to the host and merges there; here every device is driven by its own host thread, the
// partial populations merge with an RCCL all-reduce(sum, uint32) over xGMI, the neighbour partials with
// ONE all-reduce(min, uint64) of the packed (d2 bits << 32 | index) words, the Boruvka candidates of the
// screening forest with an all-reduce(min, uint64), and only the final arrays cross PCIe, once, from
// device 0.  One segment per device (dc_hip_*_segment_dev: every G-th query group of the spatial order).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first session that spans more than one
// device): a single-GPU process -- the common case of the command line, and every python process that
// already carries torch's copy of the library -- neither pays for loading the 570 MB library nor ends
// up with two copies of it.  A session that needs it and cannot load it, or whose communicator cannot be
// built, falls back to the reference's own merge: every partial is copied to the host, summed / minimised
// there and written back to every device (density_clustering_cuda.cu:171-180, :311-326) -- slower, same
// results.  DC_SESSION_MERGE=host / rccl forces one or the other (rccl: failing to get it is an error).
#include "../../include/dc_density.h"
#include "dc_common.hpp"
#include "dc_mfma.hpp"

#include <rccl/rccl.h>   // types and enums only; the functions are resolved with dlsym

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace dc {
int set_error(int code, const char* msg);   // dc_capi.hip: thread-local last error of the calling thread
}

namespace {

int failf(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return dc::set_error(code, buf);
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl();
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      x->handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (x->handle) break;
    }
    if (!x->handle) {
      x->error = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?");
      return x;
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(x->handle, name);
      if (!p && x->error.empty()) x->error = std::string("librccl: missing symbol ") + name;
      return p;
    };
    x->CommInitAll = (decltype(x->CommInitAll))sym("ncclCommInitAll");
    x->CommDestroy = (decltype(x->CommDestroy))sym("ncclCommDestroy");
    x->AllReduce = (decltype(x->AllReduce))sym("ncclAllReduce");
    x->AllGather = (decltype(x->AllGather))sym("ncclAllGather");
    x->GroupStart = (decltype(x->GroupStart))sym("ncclGroupStart");
    x->GroupEnd = (decltype(x->GroupEnd))sym("ncclGroupEnd");
    x->GetErrorString = (decltype(x->GetErrorString))sym("ncclGetErrorString");
    return x;
  }();
  return r;
}

struct DevState {
  int device = 0;
  hipStream_t stream = nullptr;
  float* d_coords = nullptr;
  void* d_ws = nullptr;
  size_t ws_bytes = 0;
  uint32_t* d_pops = nullptr;   // [pops_cap][n_rows]
  size_t pops_cap = 0;
  float* d_fe = nullptr;
  uint32_t* d_idx = nullptr;    // [2][n_rows]: nn, nn_hd
  float* d_d2 = nullptr;        // [2][n_rows]
  unsigned long long* d_words = nullptr;   // [2][n_rows] packed neighbour words / Boruvka candidates
  uint32_t* d_block = nullptr;  // [4][block_rows]: this device's neighbour block (all-gather merge)
  uint32_t* d_blocks = nullptr; // [G][4][block_rows]: the gathered blocks
  uint32_t* d_comp = nullptr;   // forest: component ids, ranks
  uint32_t* d_rank = nullptr;
  ncclComm_t comm = nullptr;
  bool stats_ready = false;     // a sweep has left the statistics of the resident coordinates in the workspace header
  int variant() const { return DC_VARIANT_AUTO | (stats_ready ? DC_FLAG_STATS_VALID : 0); }
};

}  // namespace

enum MergeMode { kMergeNone = 0, kMergeRccl = 1, kMergeHost = 2 };

struct dc_hip_session {
  size_t n_rows = 0, n_cols = 0;
  std::vector<DevState> dev;
  bool use_rccl = false;         // partials merge with RCCL collectives
  bool host_merge = false;       // ... or through the host (RCCL unavailable, or forced)
  bool have_fe = false;
  size_t n_radii = 0;            // of the populations currently resident
  uint64_t tiles_pop = 0, tiles_nn = 0;
  std::string merge_note;        // why RCCL is not used although there are several devices
  std::string merge_line;        // dc_hip_session_merge_note: the merge in one line
};

namespace {

// every entry point leaves the calling thread's current device as it found it (a torch host, or a caller
// of the _dev entry points, relies on it)
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) {
      (void)hipGetLastError();
      prev = -1;
    }
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// fn(g) on one host thread per device (density_clustering_cuda.cu:152-157, :295-299 do the same with
// OpenMP); the first failure wins and its message becomes the caller's last error
int on_every_device(dc_hip_session* s, const std::function<int(int)>& fn) {
  const int G = (int)s->dev.size();
  if (G == 1) {
    if (hipSetDevice(s->dev[0].device) != hipSuccess) return failf(DC_ERR_HIP, "hipSetDevice(%d) failed", s->dev[0].device);
    return fn(0);
  }
  std::vector<int> rc(G, DC_OK);
  std::vector<std::string> msg(G);
  std::vector<std::thread> th;
  for (int g = 0; g < G; ++g)
    th.emplace_back([&, g] {
      if (hipSetDevice(s->dev[g].device) != hipSuccess) {
        rc[g] = DC_ERR_HIP;
        msg[g] = "hipSetDevice failed";
        return;
      }
      rc[g] = fn(g);
      if (rc[g] != DC_OK) msg[g] = dc_hip_last_error();
    });
  for (auto& t : th) t.join();
  for (int g = 0; g < G; ++g)
    if (rc[g] != DC_OK) return failf(rc[g], "device %d: %s", s->dev[g].device, msg[g].c_str());
  return DC_OK;
}

#define SESSION_HIP_TRY(expr)                                                                        \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return failf(DC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// Merge of one buffer per device, in place, every device ends up with the merged array: uint32 sum (partial
// populations, density_clustering_cuda.cu:171-180) or uint64 min (packed neighbour words, :311-326; Boruvka
// candidates).  RCCL: one grouped all-reduce from this thread.  Host merge: every device's partial to the
// host (one thread per device), reduced there, the result back to every device.
int merge_partials(dc_hip_session* s, const std::function<void*(DevState&)>& buf, size_t count,
                   ncclDataType_t type, ncclRedOp_t op, const char* what) {
  const size_t G = s->dev.size();
  if (s->use_rccl) {
    Rccl* r = rccl();
    ncclResult_t e = r->GroupStart();
    for (auto& d : s->dev) {
      if (e != ncclSuccess) break;
      void* p = buf(d);
      e = r->AllReduce(p, p, count, type, op, d.comm, d.stream);
    }
    const ncclResult_t e2 = r->GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return failf(DC_ERR_HIP, "RCCL all-reduce (%s): %s", what, r->GetErrorString(e));
    return DC_OK;
  }
  if (!s->host_merge || G < 2) return DC_OK;
  const size_t esz = (type == ncclUint64) ? 8 : 4;
  std::vector<std::vector<unsigned char>> part(G);
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    part[g].resize(count * esz);
    SESSION_HIP_TRY(hipMemcpyAsync(part[g].data(), buf(d), count * esz, hipMemcpyDeviceToHost, d.stream));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
  if (rc != DC_OK) return rc;
  if (type == ncclUint64) {
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(part[0].data());
    for (size_t g = 1; g < G; ++g) {
      const unsigned long long* p = reinterpret_cast<const unsigned long long*>(part[g].data());
      for (size_t i = 0; i < count; ++i) acc[i] = (op == ncclMin) ? std::min(acc[i], p[i]) : acc[i] + p[i];
    }
  } else {
    uint32_t* acc = reinterpret_cast<uint32_t*>(part[0].data());
    for (size_t g = 1; g < G; ++g) {
      const uint32_t* p = reinterpret_cast<const uint32_t*>(part[g].data());
      for (size_t i = 0; i < count; ++i) acc[i] = (op == ncclMin) ? std::min(acc[i], p[i]) : acc[i] + p[i];
    }
  }
  return on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    SESSION_HIP_TRY(hipMemcpyAsync(buf(d), part[0].data(), count * esz, hipMemcpyHostToDevice, d.stream));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
}

// All-gather of one block of `count` uint32 per device (the neighbours of its own segment by local position,
// dc_hip_neighbors_block_pack_dev) into [G][count] on every device: what density_clustering_cuda.cu:311-326 does by
// copying row blocks on the host, and what BASELINE's north star asks of the collectives ("a final all-gather of
// nearest-neighbour indices").  RCCL: one grouped ncclAllGather; host merge: blocks to the host, all of them back.
int gather_blocks(dc_hip_session* s, size_t count) {
  const size_t G = s->dev.size();
  if (s->use_rccl) {
    Rccl* r = rccl();
    ncclResult_t e = r->GroupStart();
    for (auto& d : s->dev) {
      if (e != ncclSuccess) break;
      e = r->AllGather(d.d_block, d.d_blocks, count, ncclUint32, d.comm, d.stream);
    }
    const ncclResult_t e2 = r->GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return failf(DC_ERR_HIP, "RCCL all-gather (neighbour blocks): %s", r->GetErrorString(e));
    return DC_OK;
  }
  std::vector<uint32_t> all(G * count);
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    SESSION_HIP_TRY(hipMemcpyAsync(all.data() + (size_t)g * count, d.d_block, count * 4, hipMemcpyDeviceToHost, d.stream));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
  if (rc != DC_OK) return rc;
  return on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    SESSION_HIP_TRY(hipMemcpyAsync(d.d_blocks, all.data(), G * count * 4, hipMemcpyHostToDevice, d.stream));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
}

int sync_all(dc_hip_session* s, const char* what) {
  for (auto& d : s->dev) {
    SESSION_HIP_TRY(hipSetDevice(d.device));
    hipError_t e = hipStreamSynchronize(d.stream);
    if (e != hipSuccess) return failf(DC_ERR_HIP, "%s (device %d): %s", what, d.device, hipGetErrorString(e));
  }
  return DC_OK;
}

int read_counters(dc_hip_session* s, bool pop) {
  // evaluated 32x32 tile pairs of the sweeps that just ran, summed over the devices
  uint64_t total = 0;
  for (auto& d : s->dev) {
    if (!d.d_ws) continue;
    SESSION_HIP_TRY(hipSetDevice(d.device));
    uint64_t a = 0, b = 0;
    if (int rc = dc_hip_workspace_counters_dev(d.d_ws, &a, &b, d.stream)) return rc;
    total += pop ? a : b;
  }
  (pop ? s->tiles_pop : s->tiles_nn) = total;
  return DC_OK;
}

}  // namespace

extern "C" {

void dc_hip_session_close(dc_hip_session* s) {
  if (!s) return;
  DeviceGuard guard;
  for (auto& d : s->dev) {
    (void)hipSetDevice(d.device);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(d.comm);
    void* bufs[] = {d.d_coords, d.d_ws, d.d_pops, d.d_fe, d.d_idx, d.d_d2, d.d_words, d.d_comp, d.d_rank, d.d_block, d.d_blocks};
    for (void* p : bufs)
      if (p) (void)hipFree(p);
    if (d.stream) (void)hipStreamDestroy(d.stream);
  }
  delete s;
}

int dc_hip_session_open(const float* coords, size_t n_rows, size_t n_cols, const int* devices,
                        int n_devices, dc_hip_session** out) {
  if (!out) return failf(DC_ERR_INVALID_ARGUMENT, "null session pointer");
  *out = nullptr;
  if (!coords && n_rows) return failf(DC_ERR_INVALID_ARGUMENT, "null coords");
  if (n_cols == 0) return failf(DC_ERR_INVALID_ARGUMENT, "n_cols must be >= 1");
  if (n_cols > (size_t)dc::kMaxColsGeneric)
    return failf(DC_ERR_INVALID_ARGUMENT, "n_cols=%zu not supported (max %d)", n_cols, dc::kMaxColsGeneric);
  if (n_rows + 1 > (size_t)UINT32_MAX) return failf(DC_ERR_TOO_LARGE, "n_rows=%zu: frame ids must fit uint32", n_rows);
  const int avail = dc_hip_device_count();
  if (avail < 0) return avail;
  if (avail == 0) return failf(DC_ERR_NO_DEVICE, "no HIP device found");
  const bool caller_chose_count = n_devices > 0 && n_devices != avail;   // (a count equal to the devices present is what hosts pass for "all of them")
  if (n_devices <= 0) {
    n_devices = avail;
    devices = nullptr;
  }
  // DC_SESSION_DEVICES="0,1,..." (hosts that do not choose devices themselves -- the C++ shim, the command line): the
  // device ordinals of the session; with DC_SESSION_ALLOW_DUPLICATE_DEVICES=1 an ordinal may repeat (tests of the
  // multi-device flow on a one-GPU box).  The list never overrides a caller's choice silently: a caller that passed an
  // explicit device list keeps it, a caller that asked for N devices other than all of them gets the list only if it
  // names exactly N, and a malformed list is an error, not a shorter list.
  std::vector<int> env_devices;
  if (!devices) {
    const char* list = getenv("DC_SESSION_DEVICES");
    for (const char* c = list; c && *c;) {
      char* end = nullptr;
      const long v = strtol(c, &end, 10);
      if (end == c || (*end != ',' && *end != 0) || v < 0 || v > 4096)
        return failf(DC_ERR_INVALID_ARGUMENT, "DC_SESSION_DEVICES=\"%s\": expected a comma-separated list of device ordinals", list);
      env_devices.push_back((int)v);
      c = (*end == ',') ? end + 1 : end;
      if (*end == ',' && *c == 0)
        return failf(DC_ERR_INVALID_ARGUMENT, "DC_SESSION_DEVICES=\"%s\": trailing comma", list);
    }
    if (!env_devices.empty()) {
      if (caller_chose_count && (int)env_devices.size() != n_devices)
        return failf(DC_ERR_INVALID_ARGUMENT, "%d devices requested, but DC_SESSION_DEVICES=\"%s\" names %d: unset one of them",
                     n_devices, list, (int)env_devices.size());
      devices = env_devices.data();
      n_devices = (int)env_devices.size();
    }
  }
  {
    const char* dup = getenv("DC_SESSION_ALLOW_DUPLICATE_DEVICES");
    if (n_devices > avail && !(devices && dup && dup[0] == '1'))
      return failf(DC_ERR_INVALID_ARGUMENT, "%d devices requested, %d present", n_devices, avail);
  }
  DeviceGuard guard;
  bool duplicates = false;
  dc_hip_session* s = new dc_hip_session();
  s->n_rows = n_rows;
  s->n_cols = n_cols;
  s->dev.resize(n_devices);
  for (int g = 0; g < n_devices; ++g) {
    s->dev[g].device = devices ? devices[g] : g;
    if (s->dev[g].device < 0 || s->dev[g].device >= avail) {
      const int bad = s->dev[g].device;
      delete s;
      return failf(DC_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", bad, avail);
    }
    // DC_SESSION_ALLOW_DUPLICATE_DEVICES=1 (tests): one physical device may carry several of the session's
    // "devices" -- own stream, buffers, host thread and segment each -- so that the multi-device flow and
    // the host merge run on a one-GPU box (RCCL refuses duplicate devices: such a session merges on the host)
    for (int k = 0; k < g; ++k)
      if (s->dev[k].device == s->dev[g].device) {
        const char* dup = getenv("DC_SESSION_ALLOW_DUPLICATE_DEVICES");
        if (dup && dup[0] == '1') {
          duplicates = true;
          continue;
        }
        delete s;
        return failf(DC_ERR_INVALID_ARGUMENT, "device %d listed twice", devices[g]);
      }
  }
  const size_t bytes = sizeof(float) * n_rows * n_cols;
  // several devices pull the same host buffer at once: pin it for the duration of the uploads, so that
  // the copies are true DMA and overlap (a pageable source is staged by the calling thread)
  const bool pinned = n_devices > 1 && bytes > 0 &&
                      hipHostRegister((void*)coords, bytes, hipHostRegisterDefault) == hipSuccess;
  if (!pinned) (void)hipGetLastError();
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    SESSION_HIP_TRY(hipStreamCreate(&d.stream));
    SESSION_HIP_TRY(hipMalloc((void**)&d.d_coords, std::max<size_t>(bytes, 4)));
    if (bytes) SESSION_HIP_TRY(hipMemcpyAsync(d.d_coords, coords, bytes, hipMemcpyHostToDevice, d.stream));
    d.ws_bytes = dc_hip_workspace_bytes(n_rows, n_cols, 1);
    if (d.ws_bytes) SESSION_HIP_TRY(hipMalloc(&d.d_ws, d.ws_bytes));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
  if (pinned) (void)hipHostUnregister((void*)coords);
  // DC_SESSION_FORCE_RCCL=1: build the communicator and run the collectives on a single device as well
  // (a one-rank all-reduce is a copy onto itself): exercises the RCCL call path on a one-GPU box.
  // DC_SESSION_MERGE=host: merge through the host even where RCCL is available; =rccl: RCCL or fail.
  const char* force = getenv("DC_SESSION_FORCE_RCCL");
  const char* mode = getenv("DC_SESSION_MERGE");
  const bool want_host = (mode && strcmp(mode, "host") == 0) || duplicates;
  const bool must_rccl = mode && strcmp(mode, "rccl") == 0;
  if (rc == DC_OK && !want_host && (n_devices > 1 || (force && force[0] == '1'))) {
    Rccl* r = rccl();
    std::string why;
    if (!r->error.empty()) {
      why = r->error;
    } else {
      std::vector<ncclComm_t> comms(n_devices);
      std::vector<int> devs(n_devices);
      for (int g = 0; g < n_devices; ++g) devs[g] = s->dev[g].device;
      const ncclResult_t e = r->CommInitAll(comms.data(), n_devices, devs.data());
      if (e != ncclSuccess) {
        why = std::string("ncclCommInitAll: ") + r->GetErrorString(e);
      } else {
        for (int g = 0; g < n_devices; ++g) s->dev[g].comm = comms[g];
        s->use_rccl = true;
      }
    }
    if (!s->use_rccl && must_rccl)
      rc = failf(DC_ERR_HIP, "%d devices, DC_SESSION_MERGE=rccl: %s", n_devices, why.c_str());
    s->merge_note = why;   // (empty: RCCL is up; else why the session merges on the host)
  } else if (rc == DC_OK && n_devices > 1) {
    s->merge_note = duplicates ? "a device is listed more than once (DC_SESSION_ALLOW_DUPLICATE_DEVICES)" : "DC_SESSION_MERGE=host";
  }
  if (rc == DC_OK && n_devices > 1 && !s->use_rccl) s->host_merge = true;
  {
    char line[640];
    if (s->use_rccl)
      snprintf(line, sizeof(line), "%d device%s: partial results merge on the devices (RCCL all-reduce / all-gather over xGMI)",
               n_devices, n_devices == 1 ? "" : "s");
    else if (s->host_merge)
      snprintf(line, sizeof(line), "%d devices: partial results merge THROUGH THE HOST over PCIe, not RCCL (%s)", n_devices,
               s->merge_note.empty() ? "no reason recorded" : s->merge_note.c_str());
    else
      snprintf(line, sizeof(line), "one device: nothing to merge");
    s->merge_line = line;
  }
  if (rc != DC_OK) {
    const std::string keep = dc_hip_last_error();
    dc_hip_session_close(s);
    return dc::set_error(rc, keep.c_str());
  }
  *out = s;
  return DC_OK;
}

int dc_hip_session_devices(const dc_hip_session* s) { return s ? (int)s->dev.size() : 0; }
int dc_hip_session_uses_rccl(const dc_hip_session* s) { return (s && s->use_rccl) ? 1 : 0; }
int dc_hip_session_merge_mode(const dc_hip_session* s) {
  if (!s) return kMergeNone;
  return s->use_rccl ? kMergeRccl : (s->host_merge ? kMergeHost : kMergeNone);
}

const char* dc_hip_session_merge_note(const dc_hip_session* s) { return s ? s->merge_line.c_str() : ""; }

int dc_hip_session_counters(const dc_hip_session* s, uint64_t* pop_tiles, uint64_t* nn_tiles) {
  if (!s) return failf(DC_ERR_INVALID_ARGUMENT, "null session");
  if (pop_tiles) *pop_tiles = s->tiles_pop;
  if (nn_tiles) *nn_tiles = s->tiles_nn;
  return DC_OK;
}

int dc_hip_session_populations(dc_hip_session* s, const float* radii, size_t n_radii, uint32_t* pops) {
  DeviceGuard guard;
  if (!s || (!radii && n_radii)) return failf(DC_ERR_INVALID_ARGUMENT, "null argument");
  if (n_radii == 0 || s->n_rows == 0) return DC_OK;
  const size_t n = s->n_rows, G = s->dev.size();
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    if (d.pops_cap < n_radii) {
      if (d.d_pops) (void)hipFree(d.d_pops);
      d.d_pops = nullptr;
      d.pops_cap = 0;
      SESSION_HIP_TRY(hipMalloc((void**)&d.d_pops, sizeof(uint32_t) * n_radii * n));
      d.pops_cap = n_radii;
    }
    // (the coordinates of a session never change: after the first sweep the header statistics stay valid)
    const int r = (G == 1) ? dc_hip_populations_dev(d.d_coords, n, s->n_cols, radii, n_radii, 0, n, d.d_pops, d.d_ws,
                                                    d.ws_bytes, d.variant(), d.stream)
                           : dc_hip_populations_segment_dev(d.d_coords, n, s->n_cols, radii, n_radii, (size_t)g, G,
                                                            d.d_pops, d.d_ws, d.ws_bytes, d.variant(), d.stream);
    d.stats_ready = (r == DC_OK);
    return r;
  });
  // (whatever was resident is gone; the new populations count as resident only once merged and synchronised)
  s->n_radii = 0;
  s->have_fe = false;
  if (rc != DC_OK) return rc;
  // merge = sum of the partials (density_clustering_cuda.cu:171-180), on the devices
  if ((rc = merge_partials(s, [](DevState& d) { return (void*)d.d_pops; }, n_radii * n, ncclUint32, ncclSum,
                       "populations")) != DC_OK)
    return rc;
  if (pops) {
    SESSION_HIP_TRY(hipSetDevice(s->dev[0].device));
    SESSION_HIP_TRY(hipMemcpyAsync(pops, s->dev[0].d_pops, sizeof(uint32_t) * n_radii * n, hipMemcpyDeviceToHost,
                                   s->dev[0].stream));
  }
  if ((rc = sync_all(s, "population sweep")) != DC_OK) return rc;
  s->n_radii = n_radii;
  return read_counters(s, true);
}

int dc_hip_session_free_energies(dc_hip_session* s, size_t radius_index, float* fe, uint32_t* max_pop) {
  DeviceGuard guard;
  if (!s) return failf(DC_ERR_INVALID_ARGUMENT, "null session");
  if (s->n_rows == 0) return DC_OK;
  if (radius_index >= s->n_radii)
    return failf(DC_ERR_INVALID_ARGUMENT, "free energies of radius %zu, but %zu radii are resident", radius_index, s->n_radii);
  const size_t n = s->n_rows;
  std::vector<uint32_t> mx(s->dev.size(), 0);
  // every device computes all N values from its copy of the reduced populations (no communication)
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    if (!d.d_fe) SESSION_HIP_TRY(hipMalloc((void**)&d.d_fe, sizeof(float) * n));
    return dc_hip_free_energies_dev(d.d_pops + radius_index * n, n, d.d_fe, &mx[g], d.stream);
  });
  s->have_fe = false;
  if (rc != DC_OK) return rc;
  s->have_fe = true;   // (dc_hip_free_energies_dev synchronises its stream: the values are there)
  if (max_pop) *max_pop = mx[0];
  if (fe) {
    SESSION_HIP_TRY(hipSetDevice(s->dev[0].device));
    SESSION_HIP_TRY(hipMemcpyAsync(fe, s->dev[0].d_fe, sizeof(float) * n, hipMemcpyDeviceToHost, s->dev[0].stream));
    SESSION_HIP_TRY(hipStreamSynchronize(s->dev[0].stream));
  }
  return DC_OK;
}

int dc_hip_session_set_free_energies(dc_hip_session* s, const float* fe) {
  DeviceGuard guard;
  if (!s || !fe) return failf(DC_ERR_INVALID_ARGUMENT, "null argument");
  if (s->n_rows == 0) return DC_OK;
  const size_t n = s->n_rows;
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    if (!d.d_fe) SESSION_HIP_TRY(hipMalloc((void**)&d.d_fe, sizeof(float) * n));
    SESSION_HIP_TRY(hipMemcpyAsync(d.d_fe, fe, sizeof(float) * n, hipMemcpyHostToDevice, d.stream));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
  s->have_fe = (rc == DC_OK);
  return rc;
}

int dc_hip_session_nearest_neighbors(dc_hip_session* s, uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx,
                                     float* hd_d2, double* sigma2) {
  DeviceGuard guard;
  if (!s) return failf(DC_ERR_INVALID_ARGUMENT, "null session");
  const size_t n = s->n_rows, G = s->dev.size();
  if (n == 0) {
    if (sigma2) *sigma2 = 0.0 / 0.0;   // the reference divides by nh.size() == 0
    return DC_OK;
  }
  if (!s->have_fe) return failf(DC_ERR_INVALID_ARGUMENT, "nearest neighbours need free energies (none resident)");
  // neighbour merge: all-gather of position-ordered blocks (default), or DC_SESSION_NN_MERGE=allreduce: the all-reduce(min)
  // of packed (d2, index) words
  const char* nn_mode = getenv("DC_SESSION_NN_MERGE");
  const bool gather = !(nn_mode && strcmp(nn_mode, "allreduce") == 0);
  const size_t block_rows = dc_hip_neighbors_block_rows(n, s->n_cols, G);
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    if (!d.d_idx) SESSION_HIP_TRY(hipMalloc((void**)&d.d_idx, sizeof(uint32_t) * 2 * n));
    if (!d.d_d2) SESSION_HIP_TRY(hipMalloc((void**)&d.d_d2, sizeof(float) * 2 * n));
    int r;
    if (G == 1 && !s->use_rccl) {
      r = dc_hip_nearest_neighbors_dev(d.d_coords, n, s->n_cols, d.d_fe, 0, n, d.d_idx, d.d_d2, d.d_idx + n,
                                       d.d_d2 + n, d.d_ws, d.ws_bytes, d.variant(), d.stream);
      d.stats_ready = (r == DC_OK);
      return r;
    }
    r = dc_hip_nearest_neighbors_segment_dev(d.d_coords, n, s->n_cols, d.d_fe, (size_t)g, G, d.d_idx, d.d_d2,
                                             d.d_idx + n, d.d_d2 + n, d.d_ws, d.ws_bytes, d.variant(), d.stream);
    d.stats_ready = (r == DC_OK);
    if (r != DC_OK) return r;
    if (gather) {   // the rows of this device's segment as a dense block, by local position
      if (!d.d_block) SESSION_HIP_TRY(hipMalloc((void**)&d.d_block, sizeof(uint32_t) * 4 * block_rows));
      if (!d.d_blocks) SESSION_HIP_TRY(hipMalloc((void**)&d.d_blocks, sizeof(uint32_t) * 4 * block_rows * G));
      return dc_hip_neighbors_block_pack_dev(d.d_idx, d.d_d2, d.d_idx + n, d.d_d2 + n, n, s->n_cols, (size_t)g, G, d.d_ws,
                                             d.ws_bytes, DC_VARIANT_AUTO, d.d_block, d.stream);
    }
    if (!d.d_words) SESSION_HIP_TRY(hipMalloc((void**)&d.d_words, sizeof(unsigned long long) * 2 * n));
    return dc_hip_neighbors_pack_dev(d.d_idx, d.d_d2, d.d_idx + n, d.d_d2 + n, n, d.d_words, d.stream);
  });
  if (rc != DC_OK) return rc;
  if ((s->use_rccl || s->host_merge) && gather) {
    // ALL-GATHER of position-ordered blocks (density_clustering_cuda.cu:311-326 copies row blocks on the host): half the
    // bytes of the all-reduce(min) below and no reduction
    if ((rc = gather_blocks(s, 4 * block_rows)) != DC_OK) return rc;
    // every device must have packed under the same layout (order, deal): the G layout headers are compared BEFORE anything
    // is scattered into the result arrays -- one small copy and synchronisation on device 0 (every device holds the same
    // gathered blocks) -- so that a mismatch leaves no mis-scattered rows behind
    {
      std::vector<uint32_t> layout(G * 8);
      DevState& d0 = s->dev[0];
      SESSION_HIP_TRY(hipSetDevice(d0.device));
      for (size_t g = 0; g < G; ++g)   // (words 0..6 of the header: the last 32 entries of plane 0 of a block)
        SESSION_HIP_TRY(hipMemcpyAsync(layout.data() + 8 * g, d0.d_blocks + g * 4 * block_rows + (block_rows - 32), 32,
                                       hipMemcpyDeviceToHost, d0.stream));
      SESSION_HIP_TRY(hipStreamSynchronize(d0.stream));
      for (size_t g = 1; g < G; ++g)
        if (memcmp(layout.data(), layout.data() + 8 * g, 7 * sizeof(uint32_t)) != 0)
          return failf(DC_ERR_HIP, "neighbour blocks of devices 0 and %zu were packed under different layouts", g);
    }
    rc = on_every_device(s, [&](int g) -> int {
      DevState& d = s->dev[g];
      return dc_hip_neighbors_block_unpack_dev(d.d_blocks, n, s->n_cols, G, d.d_ws, d.ws_bytes, DC_VARIANT_AUTO, d.d_idx,
                                               d.d_d2, d.d_idx + n, d.d_d2 + n, d.stream);
    });
    if (rc != DC_OK) return rc;
  } else if (s->use_rccl || s->host_merge) {
    // DC_SESSION_NN_MERGE=allreduce: every row has one owner; all other devices hold the larger "none" word
    if ((rc = merge_partials(s, [](DevState& d) { return (void*)d.d_words; }, 2 * n, ncclUint64, ncclMin,
                         "neighbours")) != DC_OK)
      return rc;
    rc = on_every_device(s, [&](int g) -> int {
      DevState& d = s->dev[g];
      return dc_hip_neighbors_unpack_dev(d.d_words, n, d.d_idx, d.d_d2, d.d_idx + n, d.d_d2 + n, d.stream);
    });
    if (rc != DC_OK) return rc;
  }
  DevState& d0 = s->dev[0];
  SESSION_HIP_TRY(hipSetDevice(d0.device));
  std::vector<float> tmp;
  float* d2_host = nn_d2;
  if (!d2_host && sigma2) {
    tmp.resize(n);
    d2_host = tmp.data();
  }
  if (nn_idx) SESSION_HIP_TRY(hipMemcpyAsync(nn_idx, d0.d_idx, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, d0.stream));
  if (hd_idx) SESSION_HIP_TRY(hipMemcpyAsync(hd_idx, d0.d_idx + n, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, d0.stream));
  if (d2_host) SESSION_HIP_TRY(hipMemcpyAsync(d2_host, d0.d_d2, sizeof(float) * n, hipMemcpyDeviceToHost, d0.stream));
  if (hd_d2) SESSION_HIP_TRY(hipMemcpyAsync(hd_d2, d0.d_d2 + n, sizeof(float) * n, hipMemcpyDeviceToHost, d0.stream));
  if ((rc = sync_all(s, "nearest-neighbour sweep")) != DC_OK) return rc;
  if (sigma2) {
    double acc = 0.0;   // frame order, double: density_clustering.cpp:334-343
    for (size_t i = 0; i < n; ++i) acc += (double)d2_host[i];
    *sigma2 = acc / (double)n;
  }
  return read_counters(s, false);
}

int dc_hip_session_radius_pairs(dc_hip_session* s, float r2, uint32_t* pairs, size_t capacity,
                                unsigned long long* count) {
  DeviceGuard guard;
  if (!s || !count) return failf(DC_ERR_INVALID_ARGUMENT, "null argument");
  *count = 0;
  const size_t n = s->n_rows;
  if (n == 0) return DC_OK;
  if (capacity && !pairs) return failf(DC_ERR_INVALID_ARGUMENT, "null pair buffer");
  DevState& d = s->dev[0];   // the pair list is produced by one device (it is consumed on the host)
  SESSION_HIP_TRY(hipSetDevice(d.device));
  if (d.pops_cap < 1) {
    SESSION_HIP_TRY(hipMalloc((void**)&d.d_pops, sizeof(uint32_t) * n));
    d.pops_cap = 1;
  }
  s->n_radii = 0;   // (the resident populations are overwritten)
  uint32_t* d_pairs = nullptr;
  unsigned long long* d_count = nullptr;
  int rc = DC_OK;
  hipError_t e = hipMalloc((void**)&d_count, sizeof(unsigned long long));
  if (e == hipSuccess && capacity) e = hipMalloc((void**)&d_pairs, sizeof(uint32_t) * 2 * capacity);
  if (e != hipSuccess) rc = failf(DC_ERR_HIP, "radius pairs setup: %s", hipGetErrorString(e));
  if (rc == DC_OK)
    rc = dc_hip_radius_pairs_dev(d.d_coords, n, s->n_cols, r2, d.d_pops, d_pairs, capacity, d_count, d.d_ws,
                                 d.ws_bytes, d.stream);
  if (rc == DC_OK) {
    e = hipMemcpyAsync(count, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    if (e == hipSuccess && *count == ~0ull) {
      rc = failf(DC_ERR_INVALID_ARGUMENT, "radius pairs need finite coordinates");
    } else if (e == hipSuccess && capacity) {
      const size_t k = (size_t)std::min<unsigned long long>(*count, capacity);
      if (k) e = hipMemcpy(pairs, d_pairs, sizeof(uint32_t) * 2 * k, hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) rc = failf(DC_ERR_HIP, "radius pair sweep: %s", hipGetErrorString(e));
  }
  if (d_pairs) (void)hipFree(d_pairs);
  if (d_count) (void)hipFree(d_count);
  return rc;
}

int dc_hip_session_radius_forest(dc_hip_session* s, float r2, const uint32_t* rank, uint32_t* edges,
                                 size_t* n_edges, uint32_t* n_rounds) {
  DeviceGuard guard;
  if (!s || !n_edges) return failf(DC_ERR_INVALID_ARGUMENT, "null argument");
  *n_edges = 0;
  if (n_rounds) *n_rounds = 0;
  const size_t n = s->n_rows, G = s->dev.size();
  if (n <= 1) return DC_OK;
  if (!rank || !edges) return failf(DC_ERR_INVALID_ARGUMENT, "null pointer");
  // frame of every rank (and: is it a permutation?)
  std::vector<uint32_t> frame_of(n, 0xFFFFFFFFu);
  for (size_t i = 0; i < n; ++i) {
    if (rank[i] >= n || frame_of[rank[i]] != 0xFFFFFFFFu)
      return failf(DC_ERR_INVALID_ARGUMENT, "rank is not a permutation of 0..n_rows-1");
    frame_of[rank[i]] = (uint32_t)i;
  }
  s->n_radii = 0;   // (the resident populations serve as scratch below)
  int rc = on_every_device(s, [&](int g) -> int {
    DevState& d = s->dev[g];
    if (d.pops_cap < 1) {
      SESSION_HIP_TRY(hipMalloc((void**)&d.d_pops, sizeof(uint32_t) * n));
      d.pops_cap = 1;
    }
    if (!d.d_comp) SESSION_HIP_TRY(hipMalloc((void**)&d.d_comp, sizeof(uint32_t) * n));
    if (!d.d_rank) SESSION_HIP_TRY(hipMalloc((void**)&d.d_rank, sizeof(uint32_t) * n));
    if (!d.d_words) SESSION_HIP_TRY(hipMalloc((void**)&d.d_words, sizeof(unsigned long long) * 2 * n));
    SESSION_HIP_TRY(hipMemcpyAsync(d.d_rank, rank, sizeof(uint32_t) * n, hipMemcpyHostToDevice, d.stream));
    SESSION_HIP_TRY(hipStreamSynchronize(d.stream));
    return DC_OK;
  });
  if (rc != DC_OK) return rc;
  // components: union-find over frame ids, the smaller id is the root (= the component's id)
  std::vector<uint32_t> parent(n), comp(n);
  for (size_t i = 0; i < n; ++i) parent[i] = comp[i] = (uint32_t)i;
  auto find = [&](uint32_t x) {
    uint32_t root = x;
    while (parent[root] != root) root = parent[root];
    while (parent[x] != root) {
      const uint32_t next = parent[x];
      parent[x] = root;
      x = next;
    }
    return root;
  };
  std::vector<unsigned long long> best(n);
  size_t found = 0;
  uint32_t rounds = 0;
  // every round at least halves the number of components that still have a partner
  for (; rounds < 64; ++rounds) {
    uint32_t flagged = 0;
    rc = on_every_device(s, [&](int g) -> int {
      DevState& d = s->dev[g];
      SESSION_HIP_TRY(hipMemcpyAsync(d.d_comp, comp.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, d.stream));
      // (G devices: each sees the pairs of its own segment's queries; every pair is seen from both ends)
      return dc_hip_radius_min_edge_segment_dev(d.d_coords, n, s->n_cols, r2, d.d_comp, d.d_rank, (size_t)g,
                                                G > 1 ? G : 0, d.d_words, d.d_pops, d.d_ws, d.ws_bytes, d.stream);
    });
    if (rc != DC_OK) return rc;
    if ((rc = merge_partials(s, [](DevState& d) { return (void*)d.d_words; }, n, ncclUint64, ncclMin,
                         "lightest outgoing pairs")) != DC_OK)
      return rc;
    DevState& d0 = s->dev[0];
    SESSION_HIP_TRY(hipSetDevice(d0.device));
    SESSION_HIP_TRY(hipMemcpyAsync(best.data(), d0.d_words, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, d0.stream));
    if (rounds == 0) {
      uint32_t hdr[2] = {0, 0};
      SESSION_HIP_TRY(hipMemcpyAsync(hdr, d0.d_ws, sizeof(hdr), hipMemcpyDeviceToHost, d0.stream));
      SESSION_HIP_TRY(hipStreamSynchronize(d0.stream));
      flagged = hdr[1];
    }
    if ((rc = sync_all(s, "radius forest sweep")) != DC_OK) return rc;
    if (flagged != 0) return failf(DC_ERR_INVALID_ARGUMENT, "the radius graph needs finite coordinates");
    size_t joined = 0;
    for (size_t c = 0; c < n; ++c) {
      if (best[c] == ~0ull) continue;
      const uint32_t a = frame_of[(uint32_t)(best[c] >> 32)], b = frame_of[(uint32_t)best[c]];
      const uint32_t ra = find(a), rb = find(b);
      if (ra == rb) continue;   // the partner component chose the same pair
      parent[std::max(ra, rb)] = std::min(ra, rb);
      edges[2 * found] = a;
      edges[2 * found + 1] = b;
      ++found;
      ++joined;
    }
    if (joined == 0) break;
    for (size_t i = 0; i < n; ++i) comp[i] = find((uint32_t)i);
  }
  *n_edges = found;
  if (n_rounds) *n_rounds = rounds + (rounds < 64 ? 1 : 0);
  if (rounds >= 64)   // (every round at least halves the joinable components: 64 rounds cannot be needed)
    return failf(DC_ERR_HIP, "radius forest: components still merging after %u rounds (incomplete forest)", rounds);
  return DC_OK;
}

// ---- the host-pointer entry points that are one session each ---------------------------------------

int dc_hip_radius_pairs(const float* coords, size_t n_rows, size_t n_cols, float r2, int device,
                        uint32_t* pairs, size_t capacity, unsigned long long* count) {
  if (!count) return failf(DC_ERR_INVALID_ARGUMENT, "null pointer");
  *count = 0;
  if (n_rows == 0) return n_cols ? DC_OK : failf(DC_ERR_INVALID_ARGUMENT, "n_cols must be >= 1");
  dc_hip_session* s = nullptr;
  int rc = dc_hip_session_open(coords, n_rows, n_cols, &device, 1, &s);
  if (rc == DC_OK) rc = dc_hip_session_radius_pairs(s, r2, pairs, capacity, count);
  const std::string keep = rc == DC_OK ? "" : dc_hip_last_error();
  dc_hip_session_close(s);
  return rc == DC_OK ? DC_OK : dc::set_error(rc, keep.c_str());
}

int dc_hip_radius_forest(const float* coords, size_t n_rows, size_t n_cols, float r2,
                         const uint32_t* rank, int device, uint32_t* edges, size_t* n_edges,
                         uint32_t* n_rounds) {
  if (!n_edges) return failf(DC_ERR_INVALID_ARGUMENT, "null pointer");
  *n_edges = 0;
  if (n_rounds) *n_rounds = 0;
  if (n_rows <= 1) return n_cols ? DC_OK : failf(DC_ERR_INVALID_ARGUMENT, "n_cols must be >= 1");
  dc_hip_session* s = nullptr;
  int rc = dc_hip_session_open(coords, n_rows, n_cols, &device, 1, &s);
  if (rc == DC_OK) rc = dc_hip_session_radius_forest(s, r2, rank, edges, n_edges, n_rounds);
  const std::string keep = rc == DC_OK ? "" : dc_hip_last_error();
  dc_hip_session_close(s);
  return rc == DC_OK ? DC_OK : dc::set_error(rc, keep.c_str());
}

int dc_hip_density_all(const float* coords, size_t n_rows, size_t n_cols, const float* radii,
                       size_t n_radii, size_t fe_radius_index, int n_devices, uint32_t* pops,
                       float* fe, uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx, float* hd_d2) {
  if (!coords || !radii || !pops || n_radii == 0)
    return failf(DC_ERR_INVALID_ARGUMENT, "coords, radii and pops are required");
  if (fe_radius_index >= n_radii) return failf(DC_ERR_INVALID_ARGUMENT, "fe_radius_index");
  const bool want_nn = nn_idx != nullptr;
  if (want_nn && (!fe || !nn_d2 || !hd_idx || !hd_d2)) return failf(DC_ERR_INVALID_ARGUMENT, "nn outputs incomplete");
  dc_hip_session* s = nullptr;
  int rc = dc_hip_session_open(coords, n_rows, n_cols, nullptr, n_devices, &s);
  if (rc == DC_OK) rc = dc_hip_session_populations(s, radii, n_radii, pops);
  if (rc == DC_OK && (fe || want_nn)) rc = dc_hip_session_free_energies(s, fe_radius_index, fe, nullptr);
  if (rc == DC_OK && want_nn) rc = dc_hip_session_nearest_neighbors(s, nn_idx, nn_d2, hd_idx, hd_d2, nullptr);
  const std::string keep = rc == DC_OK ? "" : dc_hip_last_error();
  dc_hip_session_close(s);
  return rc == DC_OK ? DC_OK : dc::set_error(rc, keep.c_str());
}

}  // extern "C"
