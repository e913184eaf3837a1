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
// -x would set them; unset = a single rank).  The ncclUniqueId travels through a file (--id-file, default
// /tmp/pft_nccl_id.<MASTER_PORT or 0>): rank 0 writes it, the others wait for it -- one node, shared /tmp.
// With one rank the phases and collectives still run, and the result equals pft_compute()'s bit for bit (that is the
// -m gpu test; the 8-GPU run belongs to the driver's node).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <thread>

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
#define PFTOK(call)                                                                                              \
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

// rank 0 publishes the communicator id in a file (written under a temporary name, then renamed: readers never see half of it)
static bool exchange_id(ncclUniqueId* id, int rank, int world, const std::string& path) {
  if (world == 1) return ncclGetUniqueId(id) == ncclSuccess;
  if (rank == 0) {
    if (ncclGetUniqueId(id) != ncclSuccess) return false;
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(id, sizeof(*id), 1, f) == 1;
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
  }
  for (int tries = 0; tries < 6000; tries++) {  // up to a minute
    FILE* f = std::fopen(path.c_str(), "rb");
    if (f) {
      const bool ok = std::fread(id, sizeof(*id), 1, f) == 1;
      std::fclose(f);
      if (ok) return true;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  return false;
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
  if (id_file.empty()) id_file = "/tmp/pft_nccl_id." + std::to_string(env_int("MASTER_PORT", 0));

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

  const int iterations = tracker.getIterationNum();
  for (size_t f = 1; f < files.size(); f++) {
    Cloud::Ptr cloud = loadCloud(files[f]);
    if (cloud->empty()) {
      std::fprintf(stderr, "rank %d: frame %zu is empty\n", rank, f);
      return 1;
    }
    PFTOK(pft_set_input(h, cloud->points.data(), cloud->points.size()));
    PFTOK(pft_dist_begin_frame(h));  // initParticles on the first frame
    for (int it = 0; it < iterations; it++) {
      PFTOK(pft_dist_phase_a(h, it));
      NCCLOK(ncclAllReduce(d_bbox6, d_bbox6, 6, ncclFloat, ncclMax, comm, stream));
      PFTOK(pft_dist_phase_b(h));
      NCCLOK(ncclAllGather(d_shard, d_gathered, P_local * sizeof(pft_particle) / sizeof(float), ncclFloat, comm, stream));
      PFTOK(pft_dist_phase_c(h));
    }
    ParticleT result;
    PFTOK(pft_get_result(h, &result));  // synchronises the stream; also reports device-side failures of the frame
    if (rank == 0) {
      float centroid[4];
      v.objectPosition(0, result, centroid);
      printObjectLine(f, 0, result, centroid);
    }
  }
  HIPOK(hipStreamSynchronize(stream));
  NCCLOK(ncclCommDestroy(comm));
  if (rank == 0 && world > 1) ::unlink(id_file.c_str());
  hipFree(d_bbox6);
  hipFree(d_shard);
  hipFree(d_gathered);
  v.tracker_dict.clear();  // destroys the handle before the stream it enqueues on
  hipStreamDestroy(stream);
  return 0;
}
