// dist_tracking_amd.cpp -- one object's particles sharded over the GPUs of one node, host side in C++ (the reference's
// host is C++: /root/reference/src/auto_tracking.cpp:815-851), one process per GPU, RCCL over xGMI.
//
// The reference has no distributed path; SURVEY.md 8e / DESIGN.md section 6 define this one: rank r owns the global
// particle ids [r P / W, (r + 1) P / W); reference cloud, input cloud, crop and octree are replicated.  Per iteration
//
//   pft_dist_phase_a      resample the shard, pose -> matrix, AABB of the shard's transformed model
//   ncclAllReduce(max)    6 floats {-min xyz, max xyz}: the crop box is the AABB over ALL particles
//   pft_dist_phase_b      crop, octree, likelihood of the shard; raw weights into the shard buffer
//   ncclAllGather         32-byte particles with the raw weight in .weight, rank order
//   pft_dist_phase_c      normalise, weighted mean, alias table over the whole population (replicated, bit-identical)
//
// all enqueued on ONE HIP stream per rank, no host synchronisation inside a frame.  The app-level steps (model
// preparation, result consumer) are tracking_app.hpp's, as in auto_tracking_amd.cpp.
//
//   dist_tracking_amd <model> <frame0> [frame1 ...] [--particles N_TOTAL] [--seed S] [--model-leaf L] [--id-file PATH]
//
// Launch: one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK in the environment (as torch.distributed.run or mpirun
// -x would set them; unset = a single rank).  The ncclUniqueId travels through a file: --id-file PATH, or
// /tmp/pft_nccl_id.<MASTER_PORT> (several ranks need one of the two: there is no shared default path).  The file holds a
// record that a reader accepts only if it belongs to THIS launch and its publisher still runs (pft/id_exchange.hpp): a
// file left by a crashed run, or by another launch, is never taken for the id.
// Failures: a rank whose pft_* call fails keeps issuing the frame's collectives (so nobody blocks in one), and the ranks
// agree on the frame's status with a one-int all-reduce at the frame's end -- all of them stop together.  A rank waits for
// its stream with a time-out and polls ncclCommGetAsyncError; a dead peer or a failed collective ends in ncclCommAbort,
// not in a hang.
// With one rank the phases and collectives still run, and the result equals pft_compute()'s bit for bit (that is the
// -m gpu test; the 8-GPU run belongs to the driver's node).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <thread>

#include "pft/id_exchange.hpp"
#include "tracking_app.hpp"

using namespace app;

#define HIPOK(call)                                                                    \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #call, hipGetErrorString(e_)); \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)
#define NCCLOK(call)                                                                    \
  do {                                                                                  \
    ncclResult_t r_ = (call);                                                           \
    if (r_ != ncclSuccess) {                                                            \
      std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #call, ncclGetErrorString(r_)); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)
#define PFTOK(call) /* set-up only (before the first collective): a failing rank may simply leave */                                                                                             \
  do {                                                                                                           \
    int s_ = (call);                                                                                             \
    if (s_ != PFT_OK) {                                                                                          \
      std::fprintf(stderr, "rank %d: %s: %s (%s)\n", g_rank, #call, pft_status_string(s_), pft_last_error_string(h)); \
      return 1;                                                                                                  \
    }                                                                                                            \
  } while (0)

static int g_rank = 0;

static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

// rank 0 publishes the communicator id, the others wait for THIS launch's record (pft/id_exchange.hpp)
static bool exchange_id(ncclUniqueId* id, int rank, int world, const std::string& path) {
  if (world == 1) return ncclGetUniqueId(id) == ncclSuccess;
  const uint64_t nonce = pft::launch_nonce();
  if (rank == 0) {
    if (ncclGetUniqueId(id) != ncclSuccess) return false;
    return pft::publish_id(path, id, sizeof(*id), nonce);
  }
  pft::IdCheck why;
  if (pft::await_id(path, id, sizeof(*id), nonce, 60000, &why)) return true;
  std::fprintf(stderr, "rank %d: %s: %s\n", rank, path.c_str(), pft::id_check_string(why));
  return false;
}

// wait for the rank's stream without blocking forever: a peer that died or a collective that failed shows up as an
// asynchronous communicator error or as a time-out
static bool wait_stream(ncclComm_t comm, hipStream_t stream, int timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) {
      std::fprintf(stderr, "rank %d: stream: %s\n", g_rank, hipGetErrorString(q));
      return false;
    }
    ncclResult_t ar = ncclSuccess;
    if (ncclCommGetAsyncError(comm, &ar) != ncclSuccess || ar != ncclSuccess) {
      std::fprintf(stderr, "rank %d: communicator: %s\n", g_rank, ncclGetErrorString(ar));
      return false;
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
      std::fprintf(stderr, "rank %d: the frame did not finish within %d s (a peer gone?)\n", g_rank, timeout_s);
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

int main(int argc, char** argv) {
  std::vector<const char*> files;
  Options opt;
  opt.particles = 8192;
  std::string id_file;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "--model-leaf") && i + 1 < argc) opt.downsampling_grid_size = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--particles") && i + 1 < argc) opt.particles = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) opt.seed = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--id-file") && i + 1 < argc) id_file = argv[++i];
    else files.push_back(argv[i]);
  }
  if (files.size() < 2) {
    std::fprintf(stderr, "usage: %s <model> <frame>... [--particles N_TOTAL] [--seed S] [--model-leaf L] [--id-file PATH]\n", argv[0]);
    return 2;
  }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local_rank = env_int("LOCAL_RANK", rank);
  g_rank = rank;
  if (world < 1 || rank < 0 || rank >= world || opt.particles % world != 0) {
    std::fprintf(stderr, "rank %d / world %d: the particle count (%d) must divide over the ranks\n", rank, world, opt.particles);
    return 2;
  }
  if (id_file.empty()) {
    if (world > 1 && !std::getenv("MASTER_PORT")) {
      std::fprintf(stderr, "rank %d: %d ranks need a launch-unique rendezvous: set MASTER_PORT or pass --id-file PATH\n", rank, world);
      return 2;
    }
    id_file = "/tmp/pft_nccl_id." + std::to_string(env_int("MASTER_PORT", 0));
  }

  int ndev = 0;
  HIPOK(hipGetDeviceCount(&ndev));
  if (local_rank >= ndev) {
    std::fprintf(stderr, "rank %d: LOCAL_RANK %d but %d GPU(s) visible\n", rank, local_rank, ndev);
    return 1;
  }
  HIPOK(hipSetDevice(local_rank));
  hipStream_t stream;
  HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

  ncclUniqueId id;
  if (!exchange_id(&id, rank, world, id_file)) {
    std::fprintf(stderr, "rank %d: could not exchange the communicator id through %s\n", rank, id_file.c_str());
    return 1;
  }
  ncclComm_t comm;
  NCCLOK(ncclCommInitRank(&comm, world, id, rank));

  // the app: one object, its tracker sharded and on this rank's stream
  TrackingApp v(opt);
  v.ref_cloud_dict[0] = loadCloud(files[0]);
  v.buildTrackers(1, [&](ParticleFilter& tr, int) {
    tr.setDevice(local_rank);
    tr.setStream(stream);
    tr.setShard(rank, world);
  });
  if (!v.setObjectsToTrack()) return 1;
  ParticleFilter& tracker = *v.tracker_dict[0];
  if (!tracker.create()) return 1;
  pft_tracker* h = tracker.nativeHandle();

  // exchange buffers (device): bbox6, this rank's shard, the gathered population
  const size_t P = (size_t)opt.particles, P_local = P / (size_t)world;
  float* d_bbox6 = nullptr;
  pft_particle *d_shard = nullptr, *d_gathered = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void**>(&d_bbox6), 6 * sizeof(float)));
  HIPOK(hipMalloc(reinterpret_cast<void**>(&d_shard), P_local * sizeof(pft_particle)));
  HIPOK(hipMalloc(reinterpret_cast<void**>(&d_gathered), P * sizeof(pft_particle)));
  PFTOK(pft_dist_bind(h, d_bbox6, d_shard, d_gathered));

  int* d_status = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void**>(&d_status), sizeof(int)));

  // From here on the ranks are tied together by collectives.  A pft_* failure is remembered and the frame's collectives
  // are still issued; a communication failure aborts the communicator.
  bool failed = false;
  auto soft = [&](int s_, const char* what) {
    if (s_ != PFT_OK && !failed) {
      failed = true;
      std::fprintf(stderr, "rank %d: %s: %s (%s)\n", rank, what, pft_status_string(s_), pft_last_error_string(h));
    }
  };
  auto hard = [&](const char* what, const char* msg) {
    std::fprintf(stderr, "rank %d: %s: %s\n", rank, what, msg);
    ncclCommAbort(comm);
    return 1;
  };
#define COLL(call)                                                        \
  do {                                                                    \
    ncclResult_t r_ = (call);                                             \
    if (r_ != ncclSuccess) return hard(#call, ncclGetErrorString(r_));    \
  } while (0)

  const int iterations = tracker.getIterationNum();
  for (size_t f = 1; f < files.size(); f++) {
    Cloud::Ptr cloud = loadCloud(files[f]);
    if (cloud->empty()) {
      std::fprintf(stderr, "rank %d: frame %zu is empty\n", rank, f);
      failed = true;
    }
    if (!failed) soft(pft_set_input(h, cloud->points.data(), cloud->points.size()), "pft_set_input");
    if (!failed) soft(pft_dist_begin_frame(h), "pft_dist_begin_frame");  // initParticles on the first frame
    for (int it = 0; it < iterations; it++) {
      if (!failed) soft(pft_dist_phase_a(h, it), "pft_dist_phase_a");
      COLL(ncclAllReduce(d_bbox6, d_bbox6, 6, ncclFloat, ncclMax, comm, stream));
      if (!failed) soft(pft_dist_phase_b(h), "pft_dist_phase_b");
      COLL(ncclAllGather(d_shard, d_gathered, P_local * sizeof(pft_particle) / sizeof(float), ncclFloat, comm, stream));
      if (!failed) soft(pft_dist_phase_c(h), "pft_dist_phase_c");
    }
    if (!wait_stream(comm, stream, 120)) return hard("frame", "stream wait failed");
    ParticleT result;
    if (!failed) soft(pft_get_result(h, &result), "pft_get_result");  // also reports device-side failures of the frame
    // the ranks agree on the frame: 1 = fine everywhere
    int ok = failed ? 0 : 1;
    if (hipMemcpyAsync(d_status, &ok, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) return hard("status", "copy");
    COLL(ncclAllReduce(d_status, d_status, 1, ncclInt, ncclMin, comm, stream));
    if (hipMemcpyAsync(&ok, d_status, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess) return hard("status", "copy");
    if (!wait_stream(comm, stream, 120)) return hard("status", "stream wait failed");
    if (!ok) {
      if (!failed) std::fprintf(stderr, "rank %d: frame %zu failed on another rank: stopping with it\n", rank, f);
      ncclCommDestroy(comm);
      if (rank == 0 && world > 1) ::unlink(id_file.c_str());
      return 1;
    }
    if (rank == 0) {
      float centroid[4];
      v.objectPosition(0, result, centroid);
      printObjectLine(f, 0, result, centroid);
    }
  }
  HIPOK(hipStreamSynchronize(stream));
  NCCLOK(ncclCommDestroy(comm));
  if (rank == 0 && world > 1) ::unlink(id_file.c_str());
  hipFree(d_status);
  hipFree(d_bbox6);
  hipFree(d_shard);
  hipFree(d_gathered);
  v.tracker_dict.clear();  // destroys the handle before the stream it enqueues on
  hipStreamDestroy(stream);
  return 0;
}
