// ks_hip.hip — MI355X (gfx950) semantic TSDF integrator: HIP kernels + the C ABI of include/ks_hip.h.
//
// Replaces, behind the reference's plugin surface, the CPU hot path
//   kimera::FastSemanticTsdfIntegrator::integratePointCloud    [K:src/semantic_tsdf_integrator_fast.cpp:57-199]
//   kimera::MergedSemanticTsdfIntegrator::integratePointCloud  [K:src/semantic_tsdf_integrator_merged.cpp:65-329]
//   kimera::SemanticIntegratorBase::updateSemanticVoxel         [K:src/semantic_integrator_base.cpp:136-194, 283-380]
// ([K:...] = path under /root/reference/kimera_semantics/).
//
// Per frame (no CPU fallback exists):
//   stage A  points  : one lane per point — validity, T_G_C * p, start-voxel / end-voxel key  (k_points_*)
//            sort    : radix sort of point keys (start-voxel dedup slots | end-voxel bundles)   (ks_radix_sort.h)
//            rays    : exact sequential-equivalent dedup (fast) or per-bundle merge (merged)   (k_dedup / k_bundles)
//   stage B  early-out: (fast) ordered phases of k_test decide how far every ray gets (and enter its marks)        (ks_k_march.h)
//            emit    : scan of the per-position update counts, then every ray writes its (voxel, position)
//                      pairs at its own offset, allocating tiles in the spatial hash              (k_scan_local, k_emit_lane)
//            publish : pair / ray / tile counts -> pinned host memory                           (k_publish)
//   stage T  sort    : stable radix sort of the pairs on the voxel bits => every voxel's updates contiguous,
//                      in reference order (the pair list is emitted in integration order)
//            apply   : 8 lanes per voxel run — sequential TSDF + log-likelihood update, one
//                      128-byte record read and written once; runs of >= 32 updates get a workgroup each
//                      on a second stream                                                        (k_find_long, k_apply, k_apply_long)
// Kernels live in ks_k_rays.h / ks_k_march.h / ks_k_apply.h / ks_k_io.h (types: ks_types.h); this
// file is the host side: context, frame slots, the three-stream frame pipeline, the C ABI.
// Ordering contract: per voxel, updates are applied in exactly the order the reference's
// single-threaded integrator would apply them, which makes labels bit-exact.
//
// Data layout in HBM: 8x8x8-voxel tiles of 128-byte voxel records (dist | weight | colour | label
// | 21 class priors), addressed through an open-addressing hash table keyed by the packed tile index.

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstring>
#include <string.h>

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/ks_hip.h"
#include "ks_device_math.h"
#include "ks_radix_sort.h"

using namespace ksd;

#include "ks_types.h"
#include "ks_k_bundle_order.h"
#include "ks_k_rays.h"
#include "ks_k_march.h"
#include "ks_k_exact.h"
#include "ks_k_apply.h"
#include "ks_k_apply_xl.h"
#include "ks_k_shard.h"
#include "ks_k_io.h"

using namespace ksk;

namespace {

std::string g_create_error;

// Every diagnostic / A-B switch of the library sits behind ONE gate: without KS_DEBUG=1 in the environment none of them is
// read.  None changes a map (equivalent launch strategies, test sizes of buffers, traces); tests and tools that use one
// set KS_DEBUG=1 beside it.
const char* dbg_env(const char* name) {
  const char* g = getenv("KS_DEBUG");   // (read per ks_create: a test process sets and clears it between contexts)
  return (g && g[0] == '1') ? getenv(name) : nullptr;
}

}  // namespace

// ==========================================================================================
// Host side of the C ABI
// ==========================================================================================
// Everything a later stage reads from an earlier one lives in a FrameSlot.  With
// ks_config.pipeline_frames the stages of a frame run on separate streams,
//   A  points -> sort -> dedup / bundles                  (stream)
//   B  early-out phases, scan, emission, snapshot          (one of the march streams: frame % n_march)
//   T  init tiles -> sort pairs -> find_long -> apply       (stream_tail; k_apply_long beside it on stream_long, k_apply_xlong
//      on stream_xlong), enqueued by a helper thread 1..8 calls later
// so that A, B and T of neighbouring frames execute concurrently: B is a chain of small dependent launches
// (replayed as a graph captured once per slot), the sorts are bound by dependent-launch latency, the voxel
// update by memory latency, and neither A nor B touches voxel data.  The host's one wait per frame (for the
// snapshot that sizes T) never idles the GPU.  Twelve slots rotate (24 above a lag of 8); stage A of a frame waits for the tail (and
// the long runs) that last used its slot.
constexpr int kMaxLag = 16;          // largest ks_config.pipeline_frames
constexpr int kSlots = kMaxLag + 8;  // frame slots at most; a context uses ks_ctx::n_slots of them: 12 up to a lag of 8 (three batches
                                     // of four), 24 above (three batches of eight) — the tail may lag up to kMaxLag calls
constexpr int kMarchStreams = 8;
constexpr int kObsTables = 16;       // early-out tables at most; a context uses n_obs = batch x march streams of them: one per frame whose stage B
                                     // can be in flight (8 today: two batches of four, or one of eight)
struct HostSnap {
  Counters c;
  uint32_t n_tiles;
  uint32_t pad[7];
};
struct FrameSlot {
  int index = 0;
  RayDesc* d_rays = nullptr;
  float* d_deltas = nullptr;        // merged: label histograms of mixed bundles
  uint64_t* d_pairs = nullptr;      // (voxel, ray) pairs in integration order, written by k_emit
  size_t cap_pairs_in = 0;
  uint32_t* d_cnt = nullptr;        // updates per integration position (merged: 2 n entries)
  uint32_t* d_lp = nullptr;         // exclusive prefix of d_cnt inside blocks of kScanBlock
  unsigned long long* d_bt = nullptr;  // block totals of that scan
  uint8_t* d_live = nullptr;        // fast: position holds a ray that survived start-voxel dedup
  bool wide = false;                // stage B uses a whole wavefront per ray (long rays)
  FrameParams* d_F = nullptr;       // the frame's parameters in device memory (stage B reads them from there)
  uint64_t *d_gkeys = nullptr, *d_rkeys = nullptr;  // anti-grazing: this frame's sorted end-voxel keys / key per bundle
  // stage B of the batch that STARTS on this slot as captured graphs, valid for a (point count, buffers, batch size) key.
  // Exact early-out, event-driven fix point (ks_k_exact.h): THREE graphs (seed + marks + bulk rounds | finisher + commit |
  // scan + emission) with the wait for the previous frame's commit between the first two; otherwise g1 alone.
  // Two sets: [0] full batches, [1] partial ones (a flush ends a batch early: a stream that is flushed every K frames, K no
  // multiple of the batch, would otherwise re-capture on every change of the size).
  struct GraphSet {
    hipGraphExec_t g1 = nullptr, g2 = nullptr, g3 = nullptr;
    uint64_t key = 0;
  } b_graphs[2];
  // exact early-out, event-driven fix point (ks_k_exact.h): the slot's marks, per-slot table, X marks, lists
  uint64_t* d_eo_keys[2] = {nullptr, nullptr};
  uint32_t* d_eo_vals[2] = {nullptr, nullptr};
  size_t eo_cap_marks = 0;
  uint4* d_eo_tab = nullptr;
  unsigned long long* d_eo_xnode = nullptr;
  size_t eo_cap_x = 0;
  uint32_t *d_eo_cnt_b = nullptr, *d_eo_ux = nullptr, *d_eo_dirty = nullptr, *d_eo_list[2] = {nullptr, nullptr}, *d_eo_chg = nullptr,
           *d_eo_consulted = nullptr, *d_eo_lp = nullptr;
  unsigned long long* d_eo_bt = nullptr;
  EoCtl* d_eo_ctl = nullptr;
  uint32_t* d_eo_sort_ws = nullptr;
  size_t eo_sort_words = 0;
  uint32_t *d_eo_hseq = nullptr, *d_eo_where = nullptr;   // per seed mark in emission order: voxel hash | index in M
  uint4 *d_eo_rinfo = nullptr, *d_eo_ckpt = nullptr;      // per position: {u0, um, length, checkpoint step} | caster state there
  uint8_t* d_eo_hitb = nullptr;            // first iteration: per mark in emission order, the visit is a hit
  unsigned long long *d_eo_bits_a = nullptr, *d_eo_bits_b = nullptr;   // per mark of M: it counts under the current / next lengths
  unsigned long long* d_eo_btp = nullptr;  // ... exclusive prefix of the scan's block totals
  hipEvent_t eo_committed = nullptr;  // the frame's marks have entered the shared table
  Counters* d_counters = nullptr;   // inside ks_ctx::d_state
  uint32_t* d_ray_list = nullptr;   // rays to march (written by stage A, read by B)
  HostSnap* h_snap = nullptr;       // pinned + device-visible: written by k_publish at the end of B
  hipEvent_t a_done = nullptr;      // stage A complete
  hipEvent_t bl_fork = nullptr, bl_join = nullptr;   // merged: the long bundles' merge beside the bundle order (stream_bundles)
  hipEvent_t ready = nullptr;       // snapshot has landed
  hipEvent_t tail_done = nullptr;   // the tail has consumed this slot's buffers
  hipEvent_t fork = nullptr, join = nullptr;  // tail: pairs sorted and long runs listed | long runs applied
  hipEvent_t join_x = nullptr;                // the runs of more than kXLongRun updates applied (stream_xlong)
  hipEvent_t found = nullptr;                 // the long runs listed on the long-run stream (k_find_long beside k_apply_runs)
  hipEvent_t applied = nullptr;               // k_apply_runs done (stream_apply); S.join waits for it
  hipEvent_t sorted = nullptr;                // the pair sort done on the front stream (sort_on_front)
  bool tail_recorded = false;
  bool join_recorded = false;
  bool b_launched = false;    // stage B of the frame has been enqueued (with its batch)
  size_t steps_max = 0;       // longest possible ray of the frame, in voxels
  uint64_t frame_no = 0;      // the frame the slot holds
  FrameParams F{};
  size_t n = 0;
  int prof_set = -1;
  bool pending = false;
  const Counters& counters() const { return h_snap->c; }
  uint32_t n_tiles() const { return h_snap->n_tiles; }
};

// HIP-event sets for ks_profile: recorded in stream order, resolved lazily (before reuse or in
// ks_profile_get) so that profiling never adds a host wait to a frame.
constexpr int kProfSets = 32;   // 2 x the largest lag (kMaxLag = 16): a set is reused 32 frames later, long after its frame's tail
constexpr int kStageEvents = KS_STAGE_COUNT + 3;  // 0..3 stage A | 4,5 march begin/end | 6 tail begin, 7..10
struct ProfSet {
  hipEvent_t ev[kStageEvents]{};
  hipEvent_t k0 = nullptr, k1 = nullptr;  // begin/end of the k_apply dispatch itself
  bool used = false, complete = false, stages = false, apply = false, applied = false;
  uint64_t n_pairs = 0, n_points = 0;
};

struct ks_ctx {
  ks_config cfg{};
  std::string err;
  hipStream_t stream = nullptr;        // stage A (and everything else)
  // stage B; == stream unless pipelined.  Pipelined, consecutive frames march on kMarchStreams streams in
  // turn (stage B of frame i+1 does not depend on stage B of frame i: tile allocation is atomic, and the
  // early-out set of a frame is private to it when every frame bumps the set offset — then each stream
  // has its own table)
  hipStream_t stream_march_[kMarchStreams] = {};
  int n_march = 1;
  int batch = 1;                        // frames whose stage B is launched together (ks_k_march.h: BatchView)
  std::vector<FrameSlot*> batch_slots;  // frames whose stage A is enqueued and whose stage B waits for the batch to fill
  hipStream_t prof_march_stream = nullptr;  // march stream of the frame being enqueued (stage events)
  hipStream_t stream_tail = nullptr;   // stage T; == stream unless pipelined
  hipStream_t stream_long = nullptr;   // the long-run voxel update, beside k_apply (always its own stream)
  hipStream_t stream_xlong = nullptr;  // the runs of more than kXLongRun updates, beside both (k_apply_xlong)
  hipStream_t stream_bundles = nullptr; // merged, pipelined: k_bundles_long beside k_bo_* / k_bundles (round 6); == stream_long unless KS_BUNDLE_STREAM=1
  hipStream_t stream_bundles_own = nullptr;
  std::vector<hipStream_t> stream_pad;
  hipStream_t stream_apply = nullptr;  // k_apply_runs, so that the tail stream goes on with the NEXT frame's pair sort while it runs (round 6)
  bool xlong = true;
  float voxel_size_inv = 0.f, log_match = 0.f, log_non_match = 0.f;
  int vps_shift = 1;  // log2(vps / 8)

  TileTable table{};
  Pool pool{};
  uint64_t* d_start_set = nullptr;
  uint64_t* d_observed_[kObsTables] = {};
  int n_obs = 1;
  uint64_t start_offset = 0, observed_offset = 0;
  int64_t reset_counter = 0;
  uint32_t obs_tag = 0, obs_tag_lo = 1;  // frame tag of the observed set's entries (ks_k_march.h)
  Counters* d_retry_counters = nullptr;  // scratch of the pair-buffer overflow retry
  std::atomic<size_t> pairs_hint{0};     // largest pair count of a frame so far (written by the thread that runs the tails, read by the caller's)
  bool uses_early_out = false;           // fast integrator whose consecutive-collision limit can fire
  // Pipelined contexts enqueue the tail of frame i-lag on a helper thread while the calling thread enqueues
  // stages A and B of frame i (the host, not the GPU, bounds small frames: ~25 launches of ~8 us each per
  // frame).  The call still returns only after both are done, so what a call delivers does not change.
  std::thread tail_thread;
  std::mutex tail_mu;
  std::mutex capture_mu;           // a stream capture (caller) never overlaps a tail being enqueued (helper): the tail may
                                   // allocate, free or synchronise, which a capture in progress on another thread does not survive reliably
  std::condition_variable tail_cv;
  FrameSlot* tail_job = nullptr;   // posted by the caller, taken by the helper
  bool tail_busy = false, tail_quit = false;
  int tail_rc = KS_OK;
  bool use_tail_thread = false;
  double hp_a = 0, hp_b = 0, hp_t = 0, hp_sort = 0;   // KS_HOST_PROF=1: host seconds spent enqueueing stage A / B / T, radix sorts (of A+T)
  bool host_prof = false;
  bool use_graphs = true;                // stage B replayed as a hipGraph (KS_NO_GRAPH=1 or a capture failure: plain launches)
  bool test_overlap = true;              // k_test casts a long ray's next 64 voxels while the shared-set entries of the current 64 are in flight (KS_TEST_OVERLAP=0: one after the other, as measured until round 3)
  std::atomic<uint64_t> buffers_epoch{1};  // bumped whenever a buffer a captured graph points at is re-allocated
  uint8_t* d_color_lut = nullptr;   // 16 MiB rgb -> label
  uint32_t* d_label_lut = nullptr;  // 256 label -> rgba
  uint32_t tiles_initialised = 0;

  // per-frame buffers
  size_t cap_points = 0;
  float* d_xyz = nullptr;
  uint8_t* d_rgba = nullptr;
  uint8_t* d_labels = nullptr;
  uint32_t* d_hash = nullptr;
  uint32_t *d_skeys32 = nullptr, *d_skeys32b = nullptr;
  float4* d_gpw = nullptr;
  uint64_t* d_ray_keys = nullptr;
  uint2* d_glc = nullptr;
  // stage T's shared buffers exist twice (frame parity): the long runs of frame f may still be applied from
  // set f & 1 while frame f+1 sorts its pairs and lists its long runs into the other set (deferred join)
  unsigned long long* d_long_list_[2] = {nullptr, nullptr};
  uint32_t* d_blong = nullptr;
  // merged, reference bundle order: the epochs of the rehash recurrence are launched for this many bundles (the counts of the
  // frames before, with a margin; ~0: as many as the frame has points) — k_bo_rest completes a frame that has more
  std::atomic<uint32_t> bo_hint{~0u};
  bool bo_hint_fixed = false;
  uint64_t* d_key_overflow = nullptr;   // merged, compact grouping keys: the table of the end voxels outside the key window (k_points_merged)
  uint32_t key_overflow_mask = 0;
  uint32_t key_bits = 0;                // bits per axis of the key window; 0: the 64-bit keys are sorted (FrameParams::key_bits)
  float* d_blong_merged = nullptr;     // k_bundles_long -> k_bundles_long_finish: kBundleLongRec floats per long bundle
  uint64_t *d_pkeys = nullptr, *d_pkeys2 = nullptr;
  uint32_t *d_pvals = nullptr, *d_pvals2 = nullptr;
  uint32_t* d_order = nullptr;
  uint32_t* d_inv_order = nullptr;
  uint32_t *d_okeys = nullptr, *d_okeys2 = nullptr, *d_ovals = nullptr;
  size_t cap_pairs = 0;
  uint64_t* d_pairs2_[2] = {nullptr, nullptr};
  // no early-out, pipelined: stage B (scan + emission) of frames of fewer than emit_on_tail_max_pairs updates goes to the TAIL stream —
  // at 640x480 `merged` the front stream is the one that is busy all the time (stage A 0.4 ms + emission 0.09 ms per frame) and the
  // tail stream idles two thirds of it; at 1280x720 / 2 cm it is the other way round (round 6)
  bool emit_on_tail = false;
  unsigned long long emit_on_tail_max_pairs = 1ull << 23;
  bool defer_join = false;               // k_apply_long of frame f overlaps the pair sort of frame f+1 (pipelined contexts)
  hipEvent_t pending_join = nullptr;     // recorded on stream_long; the next k_apply / k_apply_long wait for it
  ksrs::Workspace sort_ws, sort_ws_tail;
  // fast, early-out in the reference's serial order (ks_k_exact.h): marks (two sets for the sort), slot ranges,
  // the reference's table content, the scan of the visited lengths, the iteration's counters
  bool exact_early_out = false;
  uint64_t* d_eo_keys[2] = {nullptr, nullptr};
  uint32_t* d_eo_vals[2] = {nullptr, nullptr};
  size_t cap_marks = 0;
  uint2* d_eo_range = nullptr;
  uint64_t* d_eo_plain = nullptr;
  uint32_t* d_eo_lp = nullptr;
  unsigned long long* d_eo_bt = nullptr;
  EoState* d_eo_state = nullptr;
  EoState* h_eo_state = nullptr;         // pinned
  uint64_t eo_iterations = 0, eo_frames = 0;  // statistics (ks_exact_early_out_stats)
  // ... event-driven (the default; KS_EXACT_HOST_LOOP=1: every frame through the host-driven loop above)
  bool eo_device = false;
  int eo_bulk_rounds = 6;                // rounds enqueued as launches before the one-workgroup finisher takes over
  int eo_sweeps = 12;                    // long rays: sweeps enqueued at a time (they end themselves once one changes nothing); the host looks
                                         // at the count once per such chunk (launch_batch) and enqueues more while rays still change
  int eo_sweep_chunks = 16;              // ... at most this many chunks, then the host-driven loop takes the frame
  int eo_sweep_order = 1;                // 0: rays in integration order; 1: a wavefront per chain segment, its rays one after the other   // long rays: one entry per emission of the marks over the rays' views = the dense iterations that follow it (ks_k_exact.h)
  std::atomic<int> eo_want_bulk{0};      // ... as a frame whose finisher was handed too long a list asks for (applied by the caller's thread between frames)
  uint32_t* d_eo_committed = nullptr;    // frames [0, *d_eo_committed) of the exact path have entered d_eo_plain
  uint32_t eo_frame_no = 0;              // frames launched through the exact path
  hipEvent_t eo_last_commit = nullptr;   // commit event of the previous frame (nullptr: nothing to wait for)
  std::atomic<size_t> eo_want_marks{0}, eo_want_x{0};   // capacities a failed frame asked for (grown by the caller's thread between frames)
  size_t eo_cap_marks = 0, eo_cap_x = 0; // per-slot capacities in use
  std::atomic<uint64_t> eo_fallbacks{0}; // frames that fell back to the host-driven loop (counted by the thread that runs the tails)
  uint64_t eo_fallbacks_seen = 0;        // ... as of the caller's last look
  std::atomic<int> eo_hopeless{0};       // consecutive frames the device loop gave up on for reasons growing a buffer does not cure
  bool eo_device_off = false;            // ... three of them: the context stays with the host-driven loop (one frame at a time)
  bool eo_trace = false;                 // KS_EXACT_TRACE=1: a line on stderr for every frame the device loop gives up (diagnostics)
  // merged in the reference's bundle order (ks_k_bundle_order.h): scratch of the rank computation, one slab
  bool use_bundle_rank = false;
  BoCtx bo{};
  BoSchedule bo_sched{};
  uint8_t* d_bo_slab = nullptr;
  int bo_epochs = 0;                     // epochs launched per frame (those a cloud of cap_points could reach)
  // ks_reduce: grow-only exchange scratch (send / receive keys and raw tile records, per-owner counts)
  int32_t* d_rx_counts = nullptr;
  size_t rx_world = 0;
  uint64_t *d_tx_keys = nullptr, *d_rx_keys = nullptr;
  uint32_t* d_tx_slots = nullptr;
  uint8_t *d_tx_payload = nullptr, *d_rx_payload = nullptr;
  size_t cap_tx = 0, cap_rx = 0;
  // scratch of the multi-GPU exchange entry points (slots / group offsets + order / distinct keys)
  uint32_t* d_xchg_u32 = nullptr;
  uint64_t* d_xchg_u64 = nullptr;
  size_t cap_xchg = 0;
  // Device words: Counters of slot k at 64 * k, the persistent tile count at 64 * kSlots.
  uint8_t* d_state = nullptr;
  FrameSlot slot[kSlots];
  int n_slots = 1;   // slots in use: 1 (unpipelined), 12 or 24 (ks_create)
  uint64_t frame_no = 0;
  ks_frame_stats owed{};  // statistics of frames completed but not yet handed to the caller (summed)
  int32_t* d_block_idx = nullptr;
  size_t cap_block_idx = 0;
  uint8_t *d_tsdf_out = nullptr, *d_sem_out = nullptr;
  uint8_t* d_vox_out = nullptr;     // staging of ks_download_updated_voxels
  size_t cap_vox_out = 0;
  size_t cap_out_blocks = 0;

  uint32_t* d_depth_blocks = nullptr;
  size_t cap_depth_blocks = 0;
  uint8_t* d_img_depth = nullptr;
  uint8_t* d_img_aux = nullptr;
  size_t cap_img_depth = 0, cap_img_aux = 0;

  int profiling = 0;  // 0 off, 1 all stages + every k_apply, 2 every 4th k_apply only
  // the voxel update of the short runs: k_apply_runs (a lane per run, runs bucketed by length: ks_k_apply.h) from this many
  // pairs per frame on, k_apply (eight lanes per voxel) below — the same records either way
  unsigned long long apply_runs_min_pairs = 1ull << 20;
  // the runs of more than kXLongRun updates through integer sums per chunk (ks_k_apply_xl.h); one set of buffers: all of it
  // runs in order on stream_xlong
  bool xl_parallel = true;
  unsigned long long xl_min_pairs = 1ull << 23;   // (below: the five launches of the path cost a 640x480 frame more than its handful of such runs through k_apply_xlong)
  XlRun* d_xl_runs = nullptr;
  XlHeader* d_xl_hdr = nullptr;
  XlChunk* d_xl_chunks = nullptr;
  uint32_t* d_xl_idx = nullptr;
  // ks_integrate_round_exact (ks_k_shard.h).  A MARCHER context: frame_tail ends with the frame's updates as records grouped by
  // owner (sh_out[*]), nothing is applied; an OWNER context: scratch for the records of one frame of a round.
  bool shard_export = false;
  int shard_world = 1;
  uint64_t shard_frames_seen = 0;        // global frames this marcher has accounted for (its own, and empty ones for the other ranks')
  uint64_t* d_sh_okey[2] = {nullptr, nullptr};
  uint64_t* d_sh_gkey[2] = {nullptr, nullptr};
  uint32_t* d_sh_seq[2] = {nullptr, nullptr};
  float* d_sh_sdf[2] = {nullptr, nullptr};
  float* d_sh_uw[2] = {nullptr, nullptr};
  uint32_t* d_sh_counts = nullptr;       // [64] per owner + [64] the origin-voxel flag
  size_t cap_sh = 0;
  uint32_t sh_counts[65] = {0};          // the last exported frame's
  uint64_t sh_exported = 0;              // its update count
  uint64_t* d_sh_tk = nullptr;           // owner: tile keys / pair keys / record numbers of the segment being applied
  uint64_t* d_sh_pairs[2] = {nullptr, nullptr};
  uint32_t* d_sh_vals[2] = {nullptr, nullptr};
  size_t cap_sh_rx = 0;
  // the runs of 33 .. 1024 updates a lane per run, bucketed by length over the frame (k_apply_long_lanes); by parity, like the lists
  bool long_lanes = true;
  unsigned long long long_lanes_min_pairs = 1ull << 24;
  bool long_min_lanes = false;
  // k_apply_runs: 256 threads (tiles of 1024 pairs) or 512 (tiles of 2048).  Measured at 1280x720 / 2 cm beside the long-run kernels:
  // 2.45 vs 3.08 ms, the frame 5.98 vs 6.26 ms (profiles/r06_c4_merged_ab.txt) — a workgroup of one wavefront per SIMD finds room
  // where one of two per SIMD does not
  uint32_t run_threads = 256u;
  uint32_t lanes_depth = 6u;   // k_apply_long_lanes: rays in flight per lane
  bool sort_on_front = false;          // the pair sort of pipelined contexts without an early-out on the front stream (frame_tail)
  FrameSlot* last_tail_of_parity[2] = {nullptr, nullptr};   // (below: too few such runs to fill wavefronts with — k_apply_long takes them all)
  unsigned long long* d_long_sorted_[2] = {nullptr, nullptr};
  LongHdr* d_long_hdr_[2] = {nullptr, nullptr};
  unsigned long long* d_xl_fb = nullptr;
  uint32_t cap_xl_chunks = 1u << 17;   // 8 M updates in such runs per frame (more: the serial kernel takes the rest)
  ks_profile prof{};
  ProfSet pset[kProfSets];
  bool fatal = false;
};

#define HIPCHK(ctx, expr)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
      return KS_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

namespace {

struct HostTimer {
  double* acc;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimer(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
  ~HostTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
inline hipStream_t march_stream(ks_ctx* c, uint64_t frame_no) { return c->stream_march_[frame_no % (uint64_t)c->n_march]; }
inline uint64_t* observed_table(ks_ctx* c, uint64_t frame_no) { return c->d_observed_[frame_no % (uint64_t)c->n_obs]; }
int sync_march(ks_ctx* c) {
  for (int i = 0; i < c->n_march; ++i)
    if (c->stream_march_[i] != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_march_[i]));
  return KS_OK;
}

template <typename T>
int dev_alloc(ks_ctx* c, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPCHK(c, hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
  return KS_OK;
}

// Rehash schedule of the host's libstdc++ unordered_map (the container the reference keeps its bundles in,
// [K:include/kimera_semantics/common.h:37]): probed from a real container, once, up to the element count needed.
void probe_rehash_schedule(size_t n_max, BoSchedule* out) {
  static std::mutex mu;
  static std::vector<uint32_t> t, b;
  static size_t probed = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (probed < n_max) {
    t.clear();
    b.clear();
    std::unordered_map<uint32_t, char> m;
    size_t bc = m.bucket_count();
    for (uint32_t i = 0; i < n_max; ++i) {
      m.emplace(i, 0);
      if (m.bucket_count() != bc) {  // inserting element i took over a new bucket array
        bc = m.bucket_count();
        t.push_back(i);
        b.push_back((uint32_t)bc);
      }
    }
    probed = n_max;
  }
  *out = BoSchedule{};
  uint32_t off = 0;
  int e = 0;
  for (; e < (int)t.size() && e < kBoMaxEpochs && t[e] < n_max; ++e) {
    out->t[e] = t[e];
    out->b[e] = b[e];
    out->head_off[e] = off;
    off += b[e];
  }
  out->n_epochs = (uint32_t)e;
  out->t[e] = UINT32_MAX;
}

int ensure_bundle_order(ks_ctx* c, size_t cap) {
  probe_rehash_schedule(cap, &c->bo_sched);
  const BoSchedule& S = c->bo_sched;
  {
    // The closed form of the iteration order (ks_k_bundle_order.h) is that of libstdc++'s unordered_map — the container the
    // reference is built with — and k_bo_small knows its tenth bucket count.  A library built against another standard
    // library (another prime policy) would compute some other, self-consistent order: refuse instead.
    static const uint32_t kLibstdcxxBuckets[] = {13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753, 42043, 85229, 172933};
    for (uint32_t e = 0; e < S.n_epochs && e < sizeof(kLibstdcxxBuckets) / sizeof(uint32_t); ++e)
      if (S.b[e] != kLibstdcxxBuckets[e] || (e > 0 && S.t[e] != kLibstdcxxBuckets[e - 1])) {
        c->err = "bundle_order = KS_BUNDLE_ORDER_REFERENCE needs libstdc++'s unordered_map rehash schedule (13, 29, 59, ...): this build's standard "
                 "library differs; use KS_BUNDLE_ORDER_CANONICAL";
        return KS_ERR_UNSUPPORTED;
      }
  }
  c->bo_epochs = (int)S.n_epochs;
  size_t heads = 0;
  for (uint32_t e = 0; e < S.n_epochs; ++e) heads += S.b[e];
  const size_t nbt = cap / kBoBlock + 2, nfbt = 2 * cap / kBoBlock + 2;
  auto al = [](size_t words) { return (words * 4 + 255) & ~(size_t)255; };
  const size_t per_map = 11 * al(cap) + al(nbt) + al(heads);
  const size_t total = 2 * per_map + al(2) + al((sizeof(BoSchedule) + 3) / 4) + 2 * al(2 * cap) + al(nfbt) + al(cap);
  if (c->d_bo_slab) (void)hipFree(c->d_bo_slab);
  c->d_bo_slab = nullptr;
  HIPCHK(c, hipMalloc((void**)&c->d_bo_slab, total));
  uint8_t* p = c->d_bo_slab;
  auto take = [&](size_t words) { uint32_t* r = (uint32_t*)p; p += al(words); return r; };
  for (int m = 0; m < 2; ++m) {
    BoMap& M = c->bo.m[m];
    M.H = take(cap); M.rank = take(cap);
    M.next[0] = take(cap); M.next[1] = take(cap);
    M.idj[0] = take(cap); M.idj[1] = take(cap);
    M.kj[0] = take(cap); M.kj[1] = take(cap);
    M.lp = take(cap); M.gm = take(cap); M.cj = take(cap);
    M.bt = take(nbt);
    M.head = take(heads);
    HIPCHK(c, hipMemsetAsync(M.head, 0xff, heads * 4, c->stream));  // chains are empty between frames
  }
  c->bo.B = take(2);
  BoSchedule* d_sched = (BoSchedule*)take((sizeof(BoSchedule) + 3) / 4);
  c->bo.sched = d_sched;
  c->bo.flag = take(2 * cap);
  c->bo.flag_lp = take(2 * cap);
  c->bo.flag_bt = take(nfbt);
  c->bo.t_of_head = take(cap);
  HIPCHK(c, hipMemcpyAsync(d_sched, &c->bo_sched, sizeof(BoSchedule), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KS_OK;
}

// chains of the ordered-phase schedule = the groups of the frame's integration order (ks_types.h: FrameParams::chains)
inline uint32_t order_chains(int order_mode, size_t n) {
  const size_t q = n / kOrderStep;
  return (order_mode == KS_ORDER_MIXED && q >= 1) ? (uint32_t)q : kOrderStep;
}
// ... and what a frame of at most `cap` points can reach: stage B's launches are sized by the slot's capacity
inline uint32_t order_chains_cap(int order_mode, size_t cap) { return std::max(order_chains(order_mode, cap), order_mode == KS_ORDER_MIXED ? kOrderStep : 0u); }
inline uint32_t order_generations_cap(int order_mode, size_t cap) {
  // KS_ORDER_MIXED: n / (n / 1024) < 2048 generations for every n >= 1024; one generation below that.
  // The other orders: 1024 chains, ceil(n / 1024) generations.
  if (order_mode == KS_ORDER_MIXED) return cap >= kOrderStep ? 2u * kOrderStep - 1u : 1u;
  return (uint32_t)((cap + kOrderStep - 1) / kOrderStep);
}
// phase boundaries (in generations: one integration position per chain) of the ordered-phase early-out
std::vector<uint32_t> phase_bounds(uint32_t n_gen, int growth) {
  std::vector<uint32_t> b{0};
  for (;;) {
    const uint64_t inc = std::max<uint64_t>(1, (uint64_t)b.back() * (uint64_t)(growth - 16) / 16);
    if (b.back() + inc >= n_gen) break;
    b.push_back((uint32_t)(b.back() + inc));
  }
  return b;
}

// longest possible ray of a frame of this context, in voxels
size_t steps_max_of(const ks_config& cfg, float voxel_size_inv) {
  const double max_len = (double)cfg.max_ray_length_m + 2.0 * (double)cfg.truncation_distance;
  return (size_t)std::ceil(1.7321 * max_len * (double)voxel_size_inv) + 8;
}
int ensure_exact_points(ks_ctx* c, size_t cap);
int ensure_points(ks_ctx* c, size_t n) {
  if (n <= c->cap_points) return KS_OK;
  // the create-time size is exact; a cloud that outgrows it gets head-room (growing completes the frames in
  // flight and re-captures stage B)
  const size_t cap = std::max<size_t>(c->cap_points ? n + n / 8 : n, 1024);
  int rc;
  if ((rc = dev_alloc(c, &c->d_xyz, cap * 3))) return rc;
  if ((rc = dev_alloc(c, &c->d_rgba, cap * 4))) return rc;
  if ((rc = dev_alloc(c, &c->d_labels, cap))) return rc;
  ++c->buffers_epoch;
  const size_t scan_cap = (c->cfg.method == KS_METHOD_MERGED ? 2 : 1) * cap;
  for (int i = 0; i < (c->cfg.pipeline_frames ? c->n_slots : 1); ++i) {
    if ((rc = dev_alloc(c, &c->slot[i].d_rays, cap))) return rc;
    if ((rc = dev_alloc(c, &c->slot[i].d_ray_list, cap))) return rc;
    if ((rc = dev_alloc(c, &c->slot[i].d_cnt, scan_cap))) return rc;
    if ((rc = dev_alloc(c, &c->slot[i].d_lp, scan_cap))) return rc;
    if ((rc = dev_alloc(c, &c->slot[i].d_bt, scan_cap / kScanBlock + 2))) return rc;
    if (c->cfg.method == KS_METHOD_FAST && (rc = dev_alloc(c, &c->slot[i].d_live, cap))) return rc;
    if (c->cfg.enable_anti_grazing && c->cfg.method == KS_METHOD_MERGED) {
      if ((rc = dev_alloc(c, &c->slot[i].d_gkeys, cap))) return rc;
      if ((rc = dev_alloc(c, &c->slot[i].d_rkeys, cap))) return rc;
    }
    if (c->cfg.method == KS_METHOD_MERGED && (rc = dev_alloc(c, &c->slot[i].d_deltas, cap * kNumLabels))) return rc;
  }
  if ((rc = dev_alloc(c, &c->d_hash, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_skeys32, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_skeys32b, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pkeys, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pkeys2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pvals, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_pvals2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_order, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_inv_order, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_okeys, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_okeys2, cap))) return rc;
  if ((rc = dev_alloc(c, &c->d_ovals, cap))) return rc;
  if (c->cfg.method == KS_METHOD_MERGED) {
    if ((rc = dev_alloc(c, &c->d_gpw, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_glc, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_ray_keys, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_blong, cap / kLongRun + 64))) return rc;
    if ((rc = dev_alloc(c, &c->d_blong_merged, (cap / kLongRun + 64) * (size_t)kBundleLongRec))) return rc;
    if (c->key_bits) {
      size_t slots = 1024;
      while (slots < 2 * cap) slots <<= 1;
      if ((rc = dev_alloc(c, &c->d_key_overflow, slots))) return rc;
      HIPCHK(c, hipMemsetAsync(c->d_key_overflow, 0, slots * sizeof(uint64_t), c->stream));
      c->key_overflow_mask = (uint32_t)(slots - 1);
    }
  }
  if (c->use_bundle_rank && (rc = ensure_bundle_order(c, cap))) return rc;
  if (c->exact_early_out) {
    if ((rc = dev_alloc(c, &c->d_eo_lp, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_eo_bt, cap / kScanBlock + 2))) return rc;
    if ((rc = ensure_exact_points(c, cap))) return rc;
  }
  c->cap_points = cap;
  return KS_OK;
}

// d_pairs is written by k_emit before the host knows the pair count: it is sized from what earlier
// frames needed (+25 %); a frame that does not fit raises kErrPairs instead of writing, and its tail
// grows the buffer and repeats the emission (frame_tail).  d_pairs2 / the long-run list are sized by
// the actual count.
int ensure_pairs_in(ks_ctx* c, FrameSlot& S, size_t bound) {
  if (bound <= S.cap_pairs_in) return KS_OK;
  const size_t cap = std::max<size_t>(bound, 1 << 20);
  int rc;
  if ((rc = dev_alloc(c, &S.d_pairs, cap))) return rc;
  S.cap_pairs_in = cap;
  ++c->buffers_epoch;  // captured stage-B graphs point at the old buffer
  return KS_OK;
}
int ensure_pairs_out(ks_ctx* c, size_t n) {
  if (n <= c->cap_pairs) return KS_OK;
  const size_t cap = std::max<size_t>(n + n / 4, 1 << 20);
  int rc;
  if (c->stream_long) HIPCHK(c, hipStreamSynchronize(c->stream_long));  // long runs of the previous frame may still read them
  if (c->sort_on_front && c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));   // (the previous frame's pair sort may still write them)
  if (c->stream_xlong) HIPCHK(c, hipStreamSynchronize(c->stream_xlong));
  if (c->stream_tail) HIPCHK(c, hipStreamSynchronize(c->stream_tail));
  for (int b = 0; b < 2; ++b) {
    if ((rc = dev_alloc(c, &c->d_pairs2_[b], cap))) return rc;
    // heads of the long runs, then (from cap / kLongRunLanes + 64 on) the heads of the runs of more than kXLongRun updates
    if ((rc = dev_alloc(c, &c->d_long_list_[b], cap / kLongRunLanes + 64 + cap / kXLongRun + 64))) return rc;
    if (c->long_lanes && (rc = dev_alloc(c, &c->d_long_sorted_[b], cap / kLongRunLanes + 64))) return rc;
  }
  if (c->d_xl_hdr && (rc = dev_alloc(c, &c->d_xl_fb, cap / kXLongRun + 64))) return rc;   // (the runs the integer-sum path leaves to k_apply_xlong)
  c->cap_pairs = cap;
  return KS_OK;
}

template <typename K>
int sort_keys(ks_ctx* c, K* a, K* b, size_t n, unsigned end_bit, K** result, unsigned begin_bit = 0, bool tail = false) {
  HostTimer ht(&c->hp_sort);
  HIPCHK(c, (ksrs::sort<K, false>(tail ? c->sort_ws_tail : c->sort_ws, a, b, nullptr, nullptr, n, end_bit,
                                  tail ? c->stream_tail : c->stream, result, nullptr, begin_bit)));
  return KS_OK;
}
template <typename K>
int sort_pairs(ks_ctx* c, K* ka, K* kb, uint32_t* va, uint32_t* vb, size_t n, unsigned end_bit, K** kres,
               uint32_t** vres) {
  HostTimer ht(&c->hp_sort);
  HIPCHK(c, (ksrs::sort<K, true>(c->sort_ws, ka, kb, va, vb, n, end_bit, c->stream, kres, vres)));
  return KS_OK;
}

inline unsigned bits_for(uint64_t n) {  // number of bits needed to represent values < n
  unsigned b = 1;
  while (b < 64 && (1ull << b) < n) ++b;
  return b;
}

int launch_batch(ks_ctx* c);
int quiesce(ks_ctx* c);
// ApproxHashSet::resetApproxSet.  `observed`: the early-out set, whose entries carry a frame tag
// (ks_k_march.h); its poison value and tag bookkeeping differ from the start-voxel set's raw hashes.
int reset_set(ks_ctx* c, uint64_t* d_set0, uint64_t* offset, bool observed) {
  const bool full = ++(*offset) >= kFullResetThreshold;
  // entries written before this offset bump can never match again; when the 10-bit frame tag is about
  // to run out they are retired in one pass and the tags start over
  const bool retag = observed && !full && c->obs_tag + (uint32_t)c->cfg.clear_checks_every_n_frames + 2u >= kObsMaxTag;
  if (full || retag) {
    // frames whose stage B still waits for its batch to fill carry the OLD tag / offset in their FrameParams: they go
    // out before the tables are rewritten (a frame with tag ~1019 running after the retag would leave marks that win
    // every atomicMax against the restarted tags 1, 2, ...)
    if (int rc = launch_batch(c)) return rc;
    if (full && c->exact_early_out) {
      // the table the exact mode keeps verbatim is rewritten below: a frame in flight whose device fix point gave up repeats
      // it on the tail stream LATER, reading and committing into that table — complete every pending tail first
      if (int rc = quiesce(c)) return rc;
    }
    if (int rc = sync_march(c)) return rc;  // stage B reads the observed set
    if (full) *offset = 0;
    for (int t = 0; t < (observed ? c->n_obs : 1); ++t) {
      uint64_t* d_set = observed ? c->d_observed_[t] : d_set0;
      if (full) {
        HIPCHK(c, hipMemsetAsync(d_set, 0, (observed ? 2 : 1) * (sizeof(uint64_t) << kSetBits), c->stream));
        const uint64_t poison = observed ? kObsPoison : ~0ull;
        HIPCHK(c, hipMemcpyAsync(d_set, &poison, sizeof(poison), hipMemcpyHostToDevice, c->stream));
      } else {
        hipLaunchKernelGGL(k_obs_retag, dim3((2u << kSetBits) / 256), dim3(256), 0, c->stream, d_set);
      }
    }
    if (observed && full && c->d_eo_plain) {  // resetApproxSet's full reset of the table the exact mode keeps verbatim
      HIPCHK(c, hipMemsetAsync(c->d_eo_plain, 0, sizeof(uint64_t) << kSetBits, c->stream));
      const uint64_t poison = ~0ull;
      HIPCHK(c, hipMemcpyAsync(c->d_eo_plain, &poison, sizeof(poison), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (observed) c->obs_tag = 0;
  }
  if (observed) c->obs_tag_lo = c->obs_tag + 1;  // the offset generation that starts with the coming frame
  return KS_OK;
}

inline void stage_mark(ks_ctx* c, int set, int ev) {
  if (set >= 0 && c->pset[set].stages)
    (void)hipEventRecord(c->pset[set].ev[ev], ev <= 3 ? c->stream : ev <= 5 ? c->prof_march_stream : c->stream_tail);
}

// fold a finished event set into ks_profile
void resolve_prof(ks_ctx* c, int set) {
  ProfSet& P = c->pset[set];
  if (!P.used || !P.complete) return;
  (void)hipEventSynchronize(P.ev[kStageEvents - 1]);
  if (P.stages) {
    for (int s = 0; s < KS_STAGE_COUNT; ++s) {
      float ms = 0.f;
      const int a = s < 3 ? s : s == 3 ? 4 : s + 2;
      if (hipEventElapsedTime(&ms, P.ev[a], P.ev[a + 1]) == hipSuccess) {
        c->prof.ms[s] += ms;
        c->prof.launches[s] += 1;
      }
    }
  }
  if (P.applied) {
    float kms = 0.f;
    if (hipEventElapsedTime(&kms, P.k0, P.k1) == hipSuccess) {
      c->prof.apply_kernel_ms += kms;
      c->prof.apply_kernel_launches += 1;
      c->prof.apply_kernel_updates += P.n_pairs;
    }
  }
  c->prof.frames += 1;
  c->prof.updates += P.n_pairs;
  c->prof.points += P.n_points;
  P.used = P.complete = P.applied = false;
}

// what stage B's kernels see of a frame slot (ks_k_march.h)
SlotView slot_view(const FrameSlot& S, Counters* counters = nullptr) {
  SlotView v{};
  v.F = S.d_F;
  v.live = S.d_live;
  v.rays = S.d_rays;
  v.cnt = S.d_cnt;
  v.lp = S.d_lp;
  v.bt = S.d_bt;
  v.ray_list = S.d_ray_list;
  v.pairs = S.d_pairs;
  v.pairs_cap = (unsigned long long)S.cap_pairs_in;
  v.C = counters ? counters : S.d_counters;
  v.host_snap = (uint32_t*)S.h_snap;
  v.eo_stats = S.d_eo_ctl ? &S.d_eo_ctl->n_x : nullptr;
  return v;
}

// pair emission over an upper bound of rays (<= n); the live ray count stays on the device
void launch_emit(ks_ctx* c, const BatchView& V, uint32_t nb, bool wide, hipStream_t st) {
  // grids and LDS are sized by the slot's capacity (the kernels take the frame's own counts from its d_F and its
  // counters): the launch sequence is the same for every frame and can be replayed
  const size_t n = c->cap_points;
  const size_t lds = ((2 * n + kScanBlock - 1) / kScanBlock) * sizeof(unsigned long long);
  if (!(c->cfg.method == KS_METHOD_MERGED && c->cfg.enable_anti_grazing)) {
    // bundles and 2 cm rays are long: 8 rays per wavefront; early-out rays are short: one per lane
    if (wide || c->cfg.method == KS_METHOD_MERGED || !c->uses_early_out) {
      hipLaunchKernelGGL(k_emit_lane<8>, dim3((uint32_t)((n + 31) / 32), nb), dim3(256), lds, st, V, c->table, c->pool);
    } else {
      hipLaunchKernelGGL(k_emit_lane<16>, dim3((uint32_t)((n + 63) / 64), nb), dim3(256), lds, st, V, c->table, c->pool);
    }
  } else if (wide) {
    hipLaunchKernelGGL(k_emit<64>, dim3((uint32_t)((n + 3) / 4), nb), dim3(256), lds, st, V, c->table, c->pool);
  } else {
    hipLaunchKernelGGL(k_emit<16>, dim3((uint32_t)((n + 15) / 16), nb), dim3(256), lds, st, V, c->table, c->pool);
  }
}

// Stage B of the nb frames of a batch, as a sequence of launches on stream sm (captured into a graph by the caller,
// or issued directly): every kernel covers the whole batch (blockIdx.y = frame).  Everything frame-specific comes
// from the slots' d_F.
// part: 0 = all of it; 1 = the early-out phases only; 2 = everything after them (the exact early-out mode runs its
// fix-point iteration, with host waits, in between)
void enqueue_stage_b(ks_ctx* c, const BatchView& V, uint32_t nb, bool wide, hipStream_t sm, size_t steps_max, int part = 0) {
  const ks_config& cfg = c->cfg;
  const size_t n = c->cap_points;  // NOT the frame's point count: see launch_emit
  if (c->uses_early_out && part != 2) {
    // ordered-phase early-out: per phase, k_test decides how far the phase's rays get against the set as it
    // stood when the phase began and enters their marks (ks_k_march.h)
    const uint32_t n_gen = order_generations_cap(cfg.integration_order_mode, n);
    const uint32_t chains_cap = order_chains_cap(cfg.integration_order_mode, n);
    const std::vector<uint32_t> B = phase_bounds(n_gen, cfg.early_out_phase_growth);
    for (size_t j = 0; j < B.size(); ++j) {
      const uint32_t g0 = B[j], g1 = (j + 1 < B.size()) ? B[j + 1] : n_gen;  // k_test ends the frame's last phase at ITS n
      const uint32_t n_sub = (g1 - g0 + kSubRun - 1) / kSubRun;  // wavefronts per chain (worst case: every ray live)
      const uint32_t steps_cap = (uint32_t)((steps_max + 3) & ~(size_t)3);
      const size_t lds_wave = (size_t)test_lds_words64(steps_cap) * sizeof(unsigned long long);
      const uint32_t wpb = lds_wave * 4 <= 60 * 1024 ? 4u : lds_wave * 2 <= 60 * 1024 ? 2u : 1u;  // wavefronts per block
      // (a frame with fewer chains than the capacity allows finds its (chain, sub-run) pairs among the first wavefronts; the rest
      // see no live ray and end)
      const dim3 grid((chains_cap * n_sub + wpb - 1) / wpb, nb), block(64 * wpb);
      if (c->test_overlap) hipLaunchKernelGGL(k_test<true>, grid, block, lds_wave * wpb, sm, V, g0, g1, steps_cap);
      else hipLaunchKernelGGL(k_test<false>, grid, block, lds_wave * wpb, sm, V, g0, g1, steps_cap);
    }
  }
  if (part == 1) return;
  if (cfg.method == KS_METHOD_MERGED && cfg.enable_anti_grazing)
    hipLaunchKernelGGL(k_count_grazing<16>, dim3((uint32_t)((n + 15) / 16), nb), dim3(256), 0, sm, V);
  hipLaunchKernelGGL(k_scan_local, dim3((uint32_t)(((cfg.method == KS_METHOD_MERGED ? 2 : 1) * n + kScanBlock - 1) / kScanBlock), nb),
                     dim3(1024), 0, sm, V);
  launch_emit(c, V, nb, wide, sm);
  // the frames' only device->host traffic: pair / ray / tile counts and error flags
  hipLaunchKernelGGL(k_publish, dim3(nb), dim3(64), 0, sm, V, (const uint32_t*)c->table.n_tiles);
}

// fast, early-out in the reference's serial order: fix-point iteration over the rays' visited lengths, seeded by
// the ordered-phase result already in S.d_cnt (ks_k_exact.h).  Host waits inside: unpipelined contexts only.
int ensure_marks(ks_ctx* c, size_t n) {
  if (n <= c->cap_marks) return KS_OK;
  const size_t cap = std::max<size_t>(n + n / 4, 1 << 20);
  int rc;
  for (int b = 0; b < 2; ++b) {
    if ((rc = dev_alloc(c, &c->d_eo_keys[b], cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_eo_vals[b], cap))) return rc;
  }
  c->cap_marks = cap;
  return KS_OK;
}
constexpr int kEoMaxIterations = 4096;
int exact_early_out(ks_ctx* c, FrameSlot& S, hipStream_t st, Counters* counters = nullptr) {
  const size_t n = S.n;
  Counters* const d_counters = counters ? counters : S.d_counters;   // (the fallback of the event-driven path runs after k_publish has cleared the slot's own)
  const FrameParams* dF = S.d_F;
  EoState* hs = c->h_eo_state;
  HIPCHK(c, hipMemsetAsync(c->d_eo_state, 0, sizeof(EoState), st));
  hipLaunchKernelGGL(k_eo_total, dim3((uint32_t)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, st, dF,
                     (const uint32_t*)S.d_cnt, c->d_eo_state);
  HIPCHK(c, hipMemcpyAsync(hs, c->d_eo_state, sizeof(EoState), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const uint32_t nb4k = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
  const size_t lds = (size_t)nb4k * sizeof(unsigned long long);
  uint64_t* kres = nullptr;
  uint32_t* vres = nullptr;
  unsigned long long n_marks = hs->n_marks_next;
  bool converged = n_marks == 0;
  int it = 0;
  for (; !converged && it < kEoMaxIterations; ++it) {
    int rc;
    if ((rc = ensure_marks(c, n_marks))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_eo_range, 0, sizeof(uint2) << kSetBits, st));
    hipLaunchKernelGGL(k_eo_scan, dim3(nb4k), dim3(1024), 0, st, dF, (const uint32_t*)S.d_cnt, c->d_eo_lp, c->d_eo_bt, c->d_eo_state);
    if (S.wide)
      hipLaunchKernelGGL(k_eo_emit<8>, dim3((uint32_t)((n + 31) / 32)), dim3(256), lds, st, dF, S.d_ray_list, S.d_rays, S.d_cnt,
                         c->d_eo_lp, c->d_eo_bt, c->d_eo_keys[0], c->d_eo_vals[0], (unsigned long long)c->cap_marks, d_counters,
                         c->d_eo_state);
    else
      hipLaunchKernelGGL(k_eo_emit<64>, dim3((uint32_t)((n + 255) / 256)), dim3(256), lds, st, dF, S.d_ray_list, S.d_rays, S.d_cnt,
                         c->d_eo_lp, c->d_eo_bt, c->d_eo_keys[0], c->d_eo_vals[0], (unsigned long long)c->cap_marks, d_counters,
                         c->d_eo_state);
    // stable sort on the slot bits only: a slot's marks stay in (position, step) order
    // (as the fallback of the event-driven path this runs on the tail's thread and stream, beside stage A of later frames:
    // the tail's sort workspace, not stage A's)
    HIPCHK(c, (ksrs::sort<uint64_t, true>(counters ? c->sort_ws_tail : c->sort_ws, c->d_eo_keys[0], c->d_eo_keys[1], c->d_eo_vals[0], c->d_eo_vals[1],
                                          (size_t)n_marks, 64, st, &kres, &vres, 44)));
    hipLaunchKernelGGL(k_eo_index, dim3((uint32_t)((n_marks + 255) / 256)), dim3(256), 0, st, n_marks, (const uint64_t*)kres,
                       c->d_eo_range);
    EoBuf E{kres, vres, c->d_eo_range, c->d_eo_plain};
    if (S.wide)
      hipLaunchKernelGGL(k_eo_eval<8>, dim3((uint32_t)((n + 31) / 32)), dim3(256), 0, st, dF, S.d_ray_list, S.d_rays, S.d_cnt, E,
                         d_counters, c->d_eo_state);
    else
      hipLaunchKernelGGL(k_eo_eval<16>, dim3((uint32_t)((n + 63) / 64)), dim3(256), 0, st, dF, S.d_ray_list, S.d_rays, S.d_cnt, E,
                         d_counters, c->d_eo_state);
    HIPCHK(c, hipMemcpyAsync(hs, c->d_eo_state, sizeof(EoState), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (hs->n_marks != n_marks) {
      c->err = "exact early-out: mark count mismatch";
      return KS_ERR_HIP;
    }
    if (hs->changed == 0) converged = true;  // the marks just sorted are the frame's marks
    else n_marks = hs->n_marks_next;
  }
  if (!converged) {
    c->err = "exact early-out: no fixed point within the iteration limit";
    return KS_ERR_UNSUPPORTED;
  }
  c->eo_iterations += (uint64_t)it;
  if (!counters) c->eo_frames += 1;   // (as the fallback the frame has been counted)
  if (n_marks)
    hipLaunchKernelGGL(k_eo_commit, dim3((uint32_t)((n_marks + 255) / 256)), dim3(256), 0, st, n_marks, (const uint64_t*)kres,
                       (const uint32_t*)vres, c->d_eo_plain);
  return KS_OK;
}

// ---- exact early-out, event-driven (ks_k_exact.h) ----------------------------------------------------------------
// Per-slot buffers: the marks are sized from what earlier frames needed (a frame that does not fit falls back to the
// host-driven loop and asks for more: eo_want_*), everything else from the slot capacity.
int ensure_exact_slots(ks_ctx* c, size_t cap_marks, size_t cap_x) {
  if (!c->eo_device) return KS_OK;
  int rc;
  const int n_slots = c->cfg.pipeline_frames ? c->n_slots : 1;
  for (int i = 0; i < n_slots; ++i) {
    FrameSlot& S = c->slot[i];
    if (!S.d_eo_ctl) {
      if ((rc = dev_alloc(c, &S.d_eo_ctl, 1))) return rc;
      if ((rc = dev_alloc(c, &S.d_eo_tab, (size_t)1 << kSetBits))) return rc;
      HIPCHK(c, hipMemsetAsync(S.d_eo_tab, 0, sizeof(uint4) << kSetBits, c->stream));
      HIPCHK(c, hipEventCreateWithFlags(&S.eo_committed, hipEventDisableTiming));
    }
    if (S.eo_cap_marks < cap_marks) {
      for (int b = 0; b < 2; ++b) {
        if ((rc = dev_alloc(c, &S.d_eo_keys[b], cap_marks))) return rc;
        if ((rc = dev_alloc(c, &S.d_eo_vals[b], cap_marks))) return rc;
      }
      S.eo_sort_words = ksrs::ws_words_dev(cap_marks, 3);
      if ((rc = dev_alloc(c, &S.d_eo_sort_ws, S.eo_sort_words))) return rc;
      if ((rc = dev_alloc(c, &S.d_eo_hitb, cap_marks))) return rc;
      if ((rc = dev_alloc(c, &S.d_eo_hseq, cap_marks))) return rc;
      if ((rc = dev_alloc(c, &S.d_eo_where, cap_marks))) return rc;
      if ((rc = dev_alloc(c, &S.d_eo_bits_a, cap_marks / 64 + 2))) return rc;
      if ((rc = dev_alloc(c, &S.d_eo_bits_b, cap_marks / 64 + 2))) return rc;
      S.eo_cap_marks = cap_marks;
    }
    if (S.eo_cap_x < cap_x) {
      if ((rc = dev_alloc(c, &S.d_eo_xnode, 2 * cap_x))) return rc;
      S.eo_cap_x = cap_x;
    }
  }
  c->eo_cap_marks = cap_marks;
  c->eo_cap_x = cap_x;
  ++c->buffers_epoch;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KS_OK;
}
int ensure_exact_points(ks_ctx* c, size_t cap) {   // the per-position arrays (called by ensure_points)
  if (!c->eo_device) return KS_OK;
  int rc;
  for (int i = 0; i < (c->cfg.pipeline_frames ? c->n_slots : 1); ++i) {
    FrameSlot& S = c->slot[i];
    for (uint32_t** p : {&S.d_eo_cnt_b, &S.d_eo_ux, &S.d_eo_dirty, &S.d_eo_list[0], &S.d_eo_list[1], &S.d_eo_chg, &S.d_eo_consulted, &S.d_eo_lp})
      if ((rc = dev_alloc(c, p, cap))) return rc;
    if ((rc = dev_alloc(c, &S.d_eo_bt, cap / kScanBlock + 2))) return rc;
    if ((rc = dev_alloc(c, &S.d_eo_btp, cap / kScanBlock + 2))) return rc;
    if ((rc = dev_alloc(c, &S.d_eo_rinfo, cap))) return rc;
    if ((rc = dev_alloc(c, &S.d_eo_ckpt, 3 * cap))) return rc;
  }
  return KS_OK;
}
EoView eo_view(ks_ctx* c, const FrameSlot& S) {
  EoView E{};
  E.F = S.d_F;
  E.ray_list = S.d_ray_list;
  E.rays = S.d_rays;
  E.C = S.d_counters;
  E.cnt_a = S.d_cnt;
  E.cnt_b = S.d_eo_cnt_b;
  E.ux = S.d_eo_ux;
  E.dirty = S.d_eo_dirty;
  E.keys = S.d_eo_keys[1];   // three radix passes leave the sorted marks in the second buffer set
  E.vals = S.d_eo_vals[1];
  E.bits_a = S.d_eo_bits_a;
  E.bits_b = S.d_eo_bits_b;
  E.lp = S.d_eo_lp;
  E.bt = S.d_eo_bt;
  E.btp = S.d_eo_btp;
  E.keys0 = S.d_eo_keys[0];
  E.vals0 = S.d_eo_vals[0];
  E.cap_marks = (unsigned long long)S.eo_cap_marks;
  E.hitb = S.d_eo_hitb;
  E.hseq_w = S.d_eo_hseq;
  E.wide = S.wide ? 1u : 0u;
  E.live = S.d_live;
  E.hseq = S.d_eo_hseq;
  E.where = S.d_eo_where;
  E.rinfo = S.d_eo_rinfo;
  E.ckpt = S.d_eo_ckpt;
  E.tab = S.d_eo_tab;
  E.xnode = S.d_eo_xnode;
  E.cap_x = (uint32_t)S.eo_cap_x;
  E.list[0] = S.d_eo_list[0];
  E.list[1] = S.d_eo_list[1];
  E.chg = S.d_eo_chg;
  E.consulted = S.d_eo_consulted;
  E.plain = c->d_eo_plain;
  E.committed = c->d_eo_committed;
  E.ctl = S.d_eo_ctl;
  return E;
}
// long rays: `count` sweeps (each ends at once when the one before it changed nothing), ks_k_exact.h
void enqueue_sweeps(ks_ctx* c, const EoBatch& Bt, uint32_t nb, int count, hipStream_t st) {
  const size_t n = c->cap_points;
  uint32_t grid;
  if (c->eo_sweep_order == 0) {
    grid = (uint32_t)std::min<size_t>((n + 3) / 4, 1 << 20);
  } else {
    const size_t waves = (size_t)order_chains_cap(c->cfg.integration_order_mode, n) *
                         ((order_generations_cap(c->cfg.integration_order_mode, n) + kEoSweepSegment - 1) / kEoSweepSegment);
    grid = (uint32_t)std::min<size_t>((waves + 3) / 4, 1 << 16);
  }
  for (int i = 0; i < count; ++i) {
    hipLaunchKernelGGL(k_eo2_sweep, dim3(grid, nb), dim3(256), 0, st, Bt, (uint32_t)c->eo_sweep_order);
    hipLaunchKernelGGL(k_eo2_sweep_next, dim3(nb), dim3(64), 0, st, Bt);
  }
}
// part 1: the seeds' marks, sorted by slot, the full first iteration and the bulk rounds — for all nb frames of the batch
// per launch (after the ordered phases, which leave the seeds in the slots' d_cnt)
int enqueue_exact_rounds(ks_ctx* c, FrameSlot* const* slots, uint32_t nb, hipStream_t st) {
  const size_t n = c->cap_points;   // (grids by capacity: the kernels read the frame's own counts)
  const uint32_t nb4k = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
  const size_t lds = (size_t)nb4k * sizeof(unsigned long long);
  EoBatch Bt{};
  ksrs::DevBatch<uint64_t> Rs{};
  const FrameSlot& S0 = *slots[0];
  for (uint32_t k = 0; k < nb; ++k) {
    const FrameSlot& S = *slots[k];
    Bt.v[k] = eo_view(c, S);
    Rs.keys_a[k] = S.d_eo_keys[0];
    Rs.keys_b[k] = S.d_eo_keys[1];
    Rs.vals_a[k] = S.d_eo_vals[0];
    Rs.vals_b[k] = S.d_eo_vals[1];
    Rs.n_dev[k] = (const unsigned long long*)&S.d_eo_ctl->st.n_marks;
    Rs.ws[k] = S.d_eo_sort_ws;
  }
  hipLaunchKernelGGL(k_eo2_begin, dim3(nb), dim3(64), 0, st, Bt);
  const uint32_t gm = (uint32_t)std::min<size_t>((S0.eo_cap_marks + 255) / 256, 2048);
  const uint32_t gn = (uint32_t)std::min<size_t>((n + 255) / 256, 2048);
  if (S0.wide) {
    // long rays: ONE emission whose views are the whole rays, then sweeps in integration order over the live bitmap, every
    // change applied at once (ks_k_exact.h); the sweeps end themselves when one of them changes nothing
    hipLaunchKernelGGL(k_eo2_full, dim3(gn, nb), dim3(256), 0, st, Bt);
    hipLaunchKernelGGL(k_eo2_scan, dim3(nb4k, nb), dim3(1024), 0, st, Bt);
    hipLaunchKernelGGL(k_eo2_emit<8>, dim3((uint32_t)((n + 31) / 32), nb), dim3(256), lds, st, Bt);
    HIPCHK(c, ksrs::sort_dev_batch<uint64_t>(Rs, (int)nb, S0.eo_sort_words, S0.eo_cap_marks, 44, 64, st));
    hipLaunchKernelGGL(k_eo2_bits, dim3(gm, nb), dim3(256), 0, st, Bt, 0u);
    hipLaunchKernelGGL(k_eo2_where, dim3(gm, nb), dim3(256), 0, st, Bt);
    enqueue_sweeps(c, Bt, nb, c->eo_sweeps, st);
    return KS_OK;
  } else {
    hipLaunchKernelGGL(k_eo2_scan, dim3(nb4k, nb), dim3(1024), 0, st, Bt);
    hipLaunchKernelGGL(k_eo2_emit<64>, dim3((uint32_t)((n + 255) / 256), nb), dim3(256), lds, st, Bt);
    // stable sort on the slot bits only: a slot's marks stay in (position, step) order; three passes: the result is in the
    // second buffer set (eo_view)
    HIPCHK(c, ksrs::sort_dev_batch<uint64_t>(Rs, (int)nb, S0.eo_sort_words, S0.eo_cap_marks, 44, 64, st));
    // the first iteration, full and streaming: hit bits of the sorted seed marks (and the slots' ranges), stop rule per ray,
    // validity bitmaps
    hipLaunchKernelGGL(k_eo2_hits, dim3(gm, nb), dim3(256), 0, st, Bt);
    hipLaunchKernelGGL(k_eo2_stop0, dim3(gn, nb), dim3(256), 0, st, Bt);
    hipLaunchKernelGGL(k_eo2_bits, dim3(gm, nb), dim3(256), 0, st, Bt, 1u);
  }
  // the event-driven rounds (round 0 was the full iteration above)
  // (wavefront per ray, grid-stride.  The lists shrink geometrically — ~3000 / 1000 / 600 / ... rays at 640x480 — and a
  // launch costs its workgroups: the later rounds get by with fewer, unless rays are long and lists stay long: 2 cm voxels)
  const uint32_t gr = (uint32_t)std::min<size_t>((n + 3) / 4, 2048);
  for (int r = 1; r <= c->eo_bulk_rounds; ++r) {
    const uint32_t g = S0.wide ? gr : std::min(gr, r == 1 ? 2048u : r == 2 ? 1024u : r <= 4 ? 512u : 256u);
    hipLaunchKernelGGL(k_eo2_eval, dim3(g, nb), dim3(256), 0, st, Bt, (uint32_t)r);
    hipLaunchKernelGGL(k_eo2_propagate, dim3(g, nb), dim3(256), 0, st, Bt, (uint32_t)r);
  }
  return KS_OK;
}
// part 2: what is left, by one workgroup, and the frame's marks into the shared table
void enqueue_exact_finish(ks_ctx* c, FrameSlot& S, hipStream_t st) {
  const EoView E = eo_view(c, S);
  hipLaunchKernelGGL(k_eo2_finish, dim3(1), dim3(kEoFinishThreads), 0, st, E, (uint32_t)c->eo_bulk_rounds + 1u, 1u);
  hipLaunchKernelGGL(k_eo2_commit, dim3(1024), dim3(256), 0, st, E);
}

// Stage B of the frames whose stage A has been enqueued (consecutive frames, at most kBatchMax): ONE sequence of
// launches for all of them, captured once per (first slot, size) and replayed.  Called by the thread that enqueues
// stage A (graph capture and the helper thread's tail never meet on a stream).
int launch_batch(ks_ctx* c) {
  if (c->batch_slots.empty()) return KS_OK;
  HostTimer htb(&c->hp_b);
  std::vector<FrameSlot*> slots;
  slots.swap(c->batch_slots);
  const uint32_t nb = (uint32_t)slots.size();
  FrameSlot& S0 = *slots[0];
  hipStream_t st = c->stream;
  hipStream_t sm = c->batch > 1 ? c->stream_march_[(S0.frame_no / (uint64_t)c->batch) % (uint64_t)c->n_march] : march_stream(c, S0.frame_no);
  const bool on_tail = c->emit_on_tail && c->profiling != 1 && (unsigned long long)c->pairs_hint.load(std::memory_order_relaxed) < c->emit_on_tail_max_pairs;
  if (on_tail) sm = c->stream_tail;
  c->prof_march_stream = sm;
  if (sm != st) HIPCHK(c, hipStreamWaitEvent(sm, slots[nb - 1]->a_done, 0));  // stage A is one in-order stream: the last frame's event covers all
  const bool stage_events = nb == 1 && S0.prof_set >= 0 && c->pset[S0.prof_set].stages;
  if (stage_events) (void)hipEventRecord(c->pset[S0.prof_set].ev[4], sm);
  // The frames' parameters go to device memory; stage B's kernels take everything else from the slots, so
  // their launch sequence depends only on the capacity: it is captured once per group of slots and replayed.
  BatchView V{};
  size_t steps_max = 0;
  ParamsBatch PB;
  for (uint32_t k = 0; k < nb; ++k) {
    FrameSlot& S = *slots[k];
    S.F.observed = observed_table(c, S.frame_no);
    if (c->exact_early_out) S.F.eo_frame = c->eo_frame_no++;
    PB.F[k] = S.F;
    PB.out[k] = S.d_F;
    V.s[k] = slot_view(S);
    steps_max = std::max(steps_max, S.steps_max);
  }
  if (nb == 1) hipLaunchKernelGGL(k_set_params, dim3(1), dim3(64), 0, sm, slots[0]->F, slots[0]->d_F);
  else hipLaunchKernelGGL(k_set_params_batch, dim3(nb), dim3(64), 0, sm, PB);   // (one launch at the head of the batch's chain instead of nb)
  const uint64_t key = ((uint64_t)c->cap_points << 24) ^ (c->buffers_epoch.load() << 4) ^ (S0.wide ? 1u : 0u) ^ ((uint64_t)nb << 1);
  FrameSlot::GraphSet& G = S0.b_graphs[nb == (uint32_t)c->batch ? 0 : 1];
  bool replayed = false;
  int rc;
  if (c->exact_early_out && c->eo_device && !c->eo_device_off) {
    // (batches of one) the ordered phases give the seed; the event-driven fix point makes it the serial result, on the
    // device: three replayed graphs, the wait for the previous frame's marks between the first two
    bool graphs = c->use_graphs && !S0.wide;   // (long rays: the host looks at the sweeps' progress between chunks of them)
    if (graphs && (G.key != key || !G.g1 || !G.g2 || !G.g3)) {
      std::lock_guard<std::mutex> cap(c->capture_mu);
      for (hipGraphExec_t* g : {&G.g1, &G.g2, &G.g3}) {
        if (*g) (void)hipGraphExecDestroy(*g);
        *g = nullptr;
      }
      G.key = 0;
      int part_rc = KS_OK;
      auto capture = [&](hipGraphExec_t* out, int part) -> bool {
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(sm, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
          if (part == 1) {
            enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max, 1);
            part_rc = enqueue_exact_rounds(c, slots.data(), nb, sm);
          } else if (part == 2) {
            for (uint32_t k = 0; k < nb; ++k) enqueue_exact_finish(c, *slots[k], sm);   // (in frame order: a frame's finisher sees the marks of the one before)
          } else {
            enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max, 2);
          }
          ok = hipStreamEndCapture(sm, &g) == hipSuccess && g != nullptr && part_rc == KS_OK;
        }
        if (ok) ok = hipGraphInstantiate(out, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        return ok;
      };
      if (capture(&G.g1, 1) && capture(&G.g2, 2) && capture(&G.g3, 3)) {
        G.key = key;
      } else {
        (void)hipGetLastError();
        for (hipGraphExec_t* g : {&G.g1, &G.g2, &G.g3}) {
          if (*g) (void)hipGraphExecDestroy(*g);
          *g = nullptr;
        }
        c->use_graphs = graphs = false;  // plain launches from now on
      }
    }
    if (graphs) HIPCHK(c, hipGraphLaunch(G.g1, sm));
    else {
      enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max, 1);
      if ((rc = enqueue_exact_rounds(c, slots.data(), nb, sm))) return rc;
      if (S0.wide) {
        // the sweeps go on until one of them changes nothing: the host reads two words per chunk of sweeps (a frame is tens of
        // milliseconds of GPU work here, and one frame at a time: ks_create)
        EoBatch Bt{};
        for (uint32_t k = 0; k < nb; ++k) Bt.v[k] = eo_view(c, *slots[k]);
        for (int chunk = 1;; ++chunk) {
          bool going = false;
          for (uint32_t k = 0; k < nb; ++k) {
            uint32_t w[2] = {0u, 0u};   // {sw_prev, fail}
            HIPCHK(c, hipMemcpyAsync(&w[0], &slots[k]->d_eo_ctl->sw_prev, sizeof(uint32_t), hipMemcpyDeviceToHost, sm));
            HIPCHK(c, hipMemcpyAsync(&w[1], &slots[k]->d_eo_ctl->fail, sizeof(uint32_t), hipMemcpyDeviceToHost, sm));
            HIPCHK(c, hipStreamSynchronize(sm));
            going = going || (w[0] != 0u && w[1] == 0u);
          }
          if (!going || chunk >= c->eo_sweep_chunks) break;
          enqueue_sweeps(c, Bt, nb, c->eo_sweeps, sm);
        }
        hipLaunchKernelGGL(k_eo2_sweep_done, dim3(nb), dim3(64), 0, sm, Bt);
      }
    }
    if (c->eo_last_commit && c->eo_last_commit != S0.eo_committed) HIPCHK(c, hipStreamWaitEvent(sm, c->eo_last_commit, 0));
    if (graphs) HIPCHK(c, hipGraphLaunch(G.g2, sm));
    else
      for (uint32_t k = 0; k < nb; ++k) enqueue_exact_finish(c, *slots[k], sm);
    HIPCHK(c, hipEventRecord(S0.eo_committed, sm));   // (after the LAST frame's commit: the batch's frames finish in order on this stream)
    c->eo_last_commit = S0.eo_committed;
    if (graphs) HIPCHK(c, hipGraphLaunch(G.g3, sm));
    else enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max, 2);
    replayed = true;
  } else if (c->exact_early_out) {
    // (batches of one) the ordered phases give the seed; the fix-point iteration (host waits inside) makes it the serial result
    enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max, 1);
    if ((rc = exact_early_out(c, S0, sm))) return rc;
    enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max, 2);
    replayed = true;
  } else if (c->use_graphs) {
    if (G.key != key || !G.g1) {
      std::lock_guard<std::mutex> cap(c->capture_mu);  // (rare: once per group of slots)
      if (G.g1) (void)hipGraphExecDestroy(G.g1);
      G.g1 = nullptr;
      G.key = 0;
      hipGraph_t g = nullptr;
      // (captured on a stream only this thread enqueues on: the tail stream is the helper thread's as well)
      hipStream_t sc = on_tail ? st : sm;
      bool ok = hipStreamBeginCapture(sc, hipStreamCaptureModeRelaxed) == hipSuccess;
      if (ok) {
        enqueue_stage_b(c, V, nb, S0.wide, sc, steps_max);
        ok = hipStreamEndCapture(sc, &g) == hipSuccess && g != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&G.g1, g, nullptr, nullptr, 0) == hipSuccess;
      if (g) (void)hipGraphDestroy(g);
      if (ok) {
        G.key = key;
      } else {
        (void)hipGetLastError();
        G.g1 = nullptr;
        c->use_graphs = false;  // plain launches from now on
      }
    }
    if (G.g1) {
      HIPCHK(c, hipGraphLaunch(G.g1, sm));
      replayed = true;
    }
  }
  if (!replayed) enqueue_stage_b(c, V, nb, S0.wide, sm, steps_max);
  for (uint32_t k = 0; k < nb; ++k) {
    HIPCHK(c, hipEventRecord(slots[k]->ready, sm));
    slots[k]->b_launched = true;
  }
  if (stage_events) (void)hipEventRecord(c->pset[S0.prof_set].ev[5], sm);
  return KS_OK;
}

// ---- front half: everything up to the counter snapshot --------------------------------------
int frame_front(ks_ctx* c, FrameSlot& S, const float Tq[7], const float* d_xyz, const uint8_t* d_rgba,
                const uint8_t* d_labels, size_t n, int freespace) {
  HostTimer hta(&c->hp_a);
  const ks_config& cfg = c->cfg;
  int rc;
  FrameParams& F = S.F;
  F = FrameParams{};
  F.T.w = Tq[0];
  F.T.v = {Tq[1], Tq[2], Tq[3]};
  F.T.t = {Tq[4], Tq[5], Tq[6]};
  F.voxel_size_inv = c->voxel_size_inv;
  F.min_ray = cfg.min_ray_length_m;
  F.max_ray = cfg.max_ray_length_m;
  F.trunc = cfg.truncation_distance;
  F.start_inv = cfg.start_voxel_subsampling_factor * c->voxel_size_inv;
  F.log_match = c->log_match;
  F.log_non_match = c->log_non_match;
  F.tsdf.voxel_size = cfg.voxel_size;
  F.tsdf.trunc = cfg.truncation_distance;
  F.tsdf.max_weight = cfg.max_weight;
  F.tsdf.dropoff_denominator = cfg.truncation_distance - cfg.voxel_size;
  F.tsdf.sparsity_factor = cfg.sparsity_compensation_factor;
  F.tsdf.use_dropoff = cfg.use_weight_dropoff;
  F.tsdf.use_sparsity = cfg.use_sparsity_compensation_factor;
  F.start_offset = c->start_offset;
  F.observed_offset = c->observed_offset;
  F.obs_tag = c->obs_tag;
  F.obs_tag_lo = c->obs_tag_lo;
  F.max_collisions = cfg.max_consecutive_ray_collisions;
  F.n = (uint32_t)n;
  {
    const uint32_t q = (uint32_t)(n / kOrderStep);
    const bool by_1024 = c->cfg.integration_order_mode == KS_ORDER_MIXED_1024_GROUPS;
    F.order_groups = q == 0 ? 1u : by_1024 ? kOrderStep : q;
    F.order_per = q == 0 ? 0u : by_1024 ? q : kOrderStep;
    F.chains = order_chains(c->cfg.integration_order_mode, n);
  }
  F.carving = cfg.voxel_carving_enabled;
  F.allow_clear = cfg.allow_clear;
  F.freespace = freespace;
  F.use_const_weight = cfg.use_const_weight;
  F.method = cfg.method;
  F.color_mode = cfg.color_mode;
  F.sorted_order = cfg.integration_order_mode == KS_ORDER_SORTED;
  F.n_dynamic = cfg.n_dynamic_labels;
  {
    const unsigned pb = bits_for(n);
    F.point_mask = (1u << pb) - 1u;
    F.clear_bit = (cfg.method == KS_METHOD_MERGED) ? (1u << pb) : 0u;
    F.seq_bits = pb + (cfg.method == KS_METHOD_MERGED ? 1u : 0u);
  }
  F.order = nullptr;  // set below once the sorted order has been computed
  F.inv_order = F.sorted_order ? c->d_inv_order : nullptr;
  std::memcpy(F.dynamic_labels, cfg.dynamic_labels, 32);
  // the early-out can never fire if the threshold exceeds the longest possible ray
  F.early_out = c->uses_early_out;

  // longest possible ray in steps: long rays get a whole wavefront per ray in stage B, short ones 16 lanes
  const size_t steps_max = steps_max_of(cfg, c->voxel_size_inv);
  const bool wide = steps_max > 400;
  S.wide = wide;
  {
    const size_t hint = c->pairs_hint.load(std::memory_order_relaxed);
    if ((rc = ensure_pairs_in(c, S, std::max<size_t>(hint + hint / 4, 1 << 20)))) return rc;
  }

  hipStream_t st = c->stream;
  if (S.tail_recorded && c->stream_tail != c->stream) HIPCHK(c, hipStreamWaitEvent(st, S.tail_done, 0));
  if (S.join_recorded) HIPCHK(c, hipStreamWaitEvent(st, S.join, 0));  // (its long runs may have ended after its tail_done)
  S.n = n;
  S.prof_set = -1;
  if (c->profiling) {
    const int set = (int)(c->frame_no % kProfSets);
    resolve_prof(c, set);
    ProfSet& P = c->pset[set];
    P.used = true;
    P.complete = P.applied = false;
    P.stages = c->profiling == 1;
    P.apply = c->profiling == 1 || (c->frame_no % 4) == 0;
    P.n_points = n;
    P.n_pairs = 0;
    S.prof_set = set;
  }
  const uint64_t this_frame = c->frame_no;
  S.frame_no = this_frame;
  ++c->frame_no;
  // S.d_counters are zero: cleared at create time / by k_publish of the slot's previous frame

  const uint32_t nb = (uint32_t)((n + 255) / 256);
  const uint32_t nb1k = (uint32_t)((n + 1023) / 1024);
  const uint32_t* order_ptr = nullptr;
  stage_mark(c, S.prof_set, 0);

  if (F.sorted_order) {
    hipLaunchKernelGGL(k_sqnorm, dim3(nb), dim3(256), 0, st, (uint32_t)n, d_xyz, c->d_okeys, c->d_ovals);
    uint32_t *ok = nullptr, *ov = nullptr;
    if ((rc = sort_pairs(c, c->d_okeys, c->d_okeys2, c->d_ovals, c->d_order, n, 32, &ok, &ov))) return rc;
    order_ptr = ov;  // position -> index
    hipLaunchKernelGGL(k_invert, dim3(nb), dim3(256), 0, st, (uint32_t)n, order_ptr, c->d_inv_order);
    F.order = order_ptr;
  }

  if (cfg.method == KS_METHOD_FAST) {
    hipLaunchKernelGGL(k_points_fast, dim3(nb1k), dim3(1024), 0, st, F, d_xyz, d_rgba, d_labels, c->d_color_lut,
                       S.d_rays, c->d_hash, c->d_skeys32, c->d_pvals, S.d_cnt, S.d_live, S.d_counters);
    stage_mark(c, S.prof_set, 1);
    // stable sort by slot only: position order inside a slot is preserved
    uint32_t *sk = nullptr, *sv = nullptr;
    if ((rc = sort_pairs(c, c->d_skeys32, c->d_skeys32b, c->d_pvals, c->d_pvals2, n, kSetBits + 1, &sk, &sv))) return rc;
    stage_mark(c, S.prof_set, 2);
    hipLaunchKernelGGL(k_dedup, dim3(nb1k), dim3(1024), 0, st, F, sk, sv, c->d_hash, c->d_start_set, S.d_ray_list,
                       S.d_rays, S.d_cnt, S.d_live, S.d_counters);
    hipLaunchKernelGGL(k_dedup_commit, dim3(nb1k), dim3(1024), 0, st, F, sk, sv, c->d_hash, c->d_start_set,
                       S.d_counters);
  } else {
    if (c->key_bits) {
      // the key window of this frame: every point within max_ray of the sensor (all but far clearing points) is inside
      const float reach = cfg.max_ray_length_m + 2.0f * cfg.voxel_size;
      const float tq[3] = {F.T.t.x, F.T.t.y, F.T.t.z};
      for (int a = 0; a < 3; ++a) {
        const float lo = std::floor((tq[a] - reach) * c->voxel_size_inv) - 2.0f;
        F.key_base[a] = (int32_t)std::min(std::max(lo, -2.0e9f), 2.0e9f);
      }
      F.key_bits = c->key_bits;
    }
    hipLaunchKernelGGL(k_points_merged, dim3(nb1k), dim3(1024), 0, st, F, d_xyz, d_rgba, d_labels, c->d_color_lut,
                       c->d_pkeys, c->d_skeys32, c->d_pvals, S.d_cnt, c->use_bundle_rank ? c->bo.flag : nullptr, c->d_key_overflow,
                       c->key_overflow_mask, S.d_counters);
    stage_mark(c, S.prof_set, 1);
    uint64_t* sk = nullptr;
    uint32_t* sk32 = nullptr;
    uint32_t* sv = nullptr;
    if (c->key_bits) {
      // four passes over 32-bit grouping keys instead of eight over the 64-bit end-voxel keys; k_gather_sorted writes the
      // sorted 64-bit keys for everything downstream
      if ((rc = sort_pairs(c, c->d_skeys32, c->d_skeys32b, c->d_pvals, c->d_pvals2, n, 32, &sk32, &sv))) return rc;
      sk = c->d_pkeys;
    } else {
      if ((rc = sort_pairs(c, c->d_pkeys, c->d_pkeys2, c->d_pvals, c->d_pvals2, n, 64, &sk, &sv))) return rc;
    }
    stage_mark(c, S.prof_set, 2);
    hipLaunchKernelGGL(k_gather_sorted, dim3(nb), dim3(256), 0, st, F, d_xyz, d_rgba, d_labels, c->d_color_lut,
                       order_ptr, (const uint64_t*)sk, (const uint32_t*)sk32, sk, sv, c->d_gpw, c->d_glc, c->use_bundle_rank ? c->bo.flag : nullptr, c->d_blong,
                       c->d_key_overflow, S.d_counters);
    // the bundles of kLongRun points and more: their merge is one serial chain per bundle (0.16 ms at 640x480 with a wall
    // close to the sensor) that needs nothing of the bundle order — pipelined, it runs on a stream of its own beside k_bo_* and
    // k_bundles, and only what needs the integration id (k_bundles_long_finish) waits for both
    const uint32_t grid_bl = (uint32_t)std::min<size_t>(n / kLongRun + 1, 2048);
    hipStream_t sb = c->stream_bundles ? c->stream_bundles : st;
    if (sb != st) {
      HIPCHK(c, hipEventRecord(S.bl_fork, st));
      HIPCHK(c, hipStreamWaitEvent(sb, S.bl_fork, 0));
      hipLaunchKernelGGL(k_bundles_long, dim3(grid_bl), dim3(128), 0, sb, F, sk, c->d_gpw, c->d_glc, c->d_blong, c->d_blong_merged, S.d_counters);
      HIPCHK(c, hipEventRecord(S.bl_join, sb));
    }
    if (c->use_bundle_rank) {
      // the bundles' ranks in the iteration order of the reference's unordered_map (ks_k_bundle_order.h):
      // insertion indices, then one walk + link launch per rehash epoch the slot capacity can reach
      const uint32_t capb = (uint32_t)((c->cap_points + kBoBlock - 1) / kBoBlock);
      const size_t lds_f = (2 * c->cap_points / kBoBlock + 3) * sizeof(uint32_t), lds_e = ((size_t)capb + 2) * sizeof(uint32_t);
      hipLaunchKernelGGL(k_bo_scan_flags, dim3((uint32_t)((2 * n + kBoBlock - 1) / kBoBlock)), dim3(kBoBlock), 0, st, (uint32_t)n, c->bo);
      hipLaunchKernelGGL(k_bo_init, dim3((uint32_t)((n + kBoBlock - 1) / kBoBlock)), dim3(kBoBlock), lds_f, st, (uint32_t)n,
                         (const uint64_t*)sk, (const uint32_t*)sv, c->bo);
      const uint32_t nbb = (uint32_t)((n + kBoBlock - 1) / kBoBlock);  // a frame of n points has at most n bundles
      // epochs whose bucket count fits one workgroup's LDS: one launch for all of them (both maps)
      int e_small = 0;
      while (e_small < c->bo_epochs && c->bo_sched.b[e_small] <= kBoSmallBuckets) ++e_small;
      if (e_small > 0) hipLaunchKernelGGL(k_bo_small, dim3(2), dim3(kBoBlock), 0, st, c->bo, e_small);
      // (at least the first launch pair's k_bo_link goes out: it takes over from k_bo_small)
      const size_t nh_floor = c->bo_hint_fixed ? (e_small > 0 ? (size_t)c->bo_sched.t[e_small - 1] + 1 : 1) : 2 * (size_t)kBoSmallBuckets;
      const size_t nh = std::min<size_t>(n, std::max<size_t>(c->bo_hint.load(std::memory_order_relaxed), nh_floor));
      int e_last = -1;   // the epoch whose k_bo_link went out without its k_bo_walk
      for (int e = e_small; e <= c->bo_epochs; ++e) {
        if (e > 0 && c->bo_sched.t[e - 1] >= nh) break;  // no map of nh bundles reaches epoch e - 1
        hipLaunchKernelGGL(k_bo_link, dim3(nbb, 2), dim3(kBoBlock), lds_e, st, c->bo, e, (e == e_small && e_small > 0) ? 1 : 0);
        if (e < c->bo_epochs && c->bo_sched.t[e] < nh)
          hipLaunchKernelGGL(k_bo_walk, dim3(nbb, 2), dim3(kBoBlock), 0, st, c->bo, e);
        else
          e_last = e;
      }
      // (a map with more bundles than the hint: the rest of the recurrence, one workgroup per map)
      if (nh < n && e_last >= 0 && e_last < c->bo_epochs) hipLaunchKernelGGL(k_bo_rest, dim3(2), dim3(kBoBlock), lds_e, st, c->bo, e_last);
    }
    // anti-grazing: the frame keeps its own copy of the keys (the next frame's stage A reuses the sort
    // buffers while this frame's emission — or its repetition after a pair-buffer overflow — may still run)
    uint64_t* ray_keys = cfg.enable_anti_grazing ? S.d_rkeys : nullptr;
    if (cfg.enable_anti_grazing) HIPCHK(c, hipMemcpyAsync(S.d_gkeys, sk, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    if (sb != st) {
      hipLaunchKernelGGL(k_bundles, dim3(nb), dim3(256), 0, st, F, sk, sv, c->d_gpw, c->d_glc, S.d_rays, S.d_deltas,
                         S.d_ray_list, ray_keys, S.d_cnt, c->bo, c->use_bundle_rank, S.d_counters);
      HIPCHK(c, hipStreamWaitEvent(st, S.bl_join, 0));
      hipLaunchKernelGGL(k_bundles_long_finish, dim3(std::min<uint32_t>((grid_bl + 3) / 4, 64)), dim3(256), 0, st, F, sk, sv, c->d_blong, c->d_blong_merged,
                         S.d_rays, S.d_deltas, S.d_ray_list, ray_keys, S.d_cnt, c->bo, c->use_bundle_rank, S.d_counters);
    } else {
      // one launch: the long bundles' serial chains in the first workgroups, the short bundles under them
      const uint32_t n_long_blocks = std::min<uint32_t>(grid_bl, 512);
      hipLaunchKernelGGL(k_bundles_all, dim3(n_long_blocks + (uint32_t)((n + 127) / 128)), dim3(128), 0, st, F, sk, sv, c->d_gpw, c->d_glc, c->d_blong,
                         S.d_rays, S.d_deltas, S.d_ray_list, ray_keys, S.d_cnt, c->bo, c->use_bundle_rank, S.d_counters, n_long_blocks);
    }
    if (cfg.enable_anti_grazing) {
      F.grazing_keys = S.d_gkeys;
      F.ray_keys = S.d_rkeys;
    }
  }
  stage_mark(c, S.prof_set, 3);
  // ---- stage B (early-out phases, scan, pair emission) is enqueued per BATCH of frames: launch_batch
  if (c->stream_march_[0] != st || c->emit_on_tail) HIPCHK(c, hipEventRecord(S.a_done, st));
  S.steps_max = steps_max;
  S.b_launched = false;
  S.pending = true;
  c->batch_slots.push_back(&S);
  if ((int)c->batch_slots.size() >= c->batch || c->profiling == 1) return launch_batch(c);
  return KS_OK;
}

// ---- tail half: sized by the snapshot --------------------------------------------------------
// Marcher: records of the frame in S (ks_k_shard.h), stable-partitioned by owner into d_sh_*[1]; counts on the host.
int shard_export_frame(ks_ctx* c, FrameSlot& S, unsigned long long n_pairs, hipStream_t st) {
  const int world = c->shard_world;
  int rc;
  if (n_pairs > c->cap_sh) {
    const size_t cap = std::max<size_t>(n_pairs + n_pairs / 4, 1 << 20);
    for (int b = 0; b < 2; ++b) {
      if ((rc = dev_alloc(c, &c->d_sh_okey[b], cap))) return rc;
      if ((rc = dev_alloc(c, &c->d_sh_gkey[b], cap))) return rc;
      if ((rc = dev_alloc(c, &c->d_sh_seq[b], cap))) return rc;
      if ((rc = dev_alloc(c, &c->d_sh_sdf[b], cap))) return rc;
      if ((rc = dev_alloc(c, &c->d_sh_uw[b], cap))) return rc;
    }
    c->cap_sh = cap;
  }
  if (!c->d_sh_counts && (rc = dev_alloc(c, &c->d_sh_counts, 128))) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_sh_counts, 0, 128 * sizeof(uint32_t), st));
  std::memset(c->sh_counts, 0, sizeof(c->sh_counts));
  c->sh_exported = n_pairs;
  if (n_pairs) {
    const uint32_t nb = (uint32_t)((n_pairs + 255) / 256);
    hipLaunchKernelGGL(k_shard_export, dim3(nb), dim3(256), 0, st, S.F, n_pairs, (const uint64_t*)S.d_pairs, (const RayDesc*)S.d_rays,
                       (const uint64_t*)c->table.slot_keys, (uint32_t)world, c->d_sh_okey[0], c->d_sh_gkey[0], c->d_sh_seq[0],
                       c->d_sh_sdf[0], c->d_sh_uw[0], c->d_sh_counts);
    uint64_t* sorted = nullptr;
    HIPCHK(c, (ksrs::sort<uint64_t, false>(c->sort_ws_tail, c->d_sh_okey[0], c->d_sh_okey[1], nullptr, nullptr, (size_t)n_pairs, 64, st,
                                           &sorted, nullptr, 56)));
    hipLaunchKernelGGL(k_shard_gather, dim3(nb), dim3(256), 0, st, n_pairs, (const uint64_t*)sorted, (const uint64_t*)c->d_sh_gkey[0],
                       (const uint32_t*)c->d_sh_seq[0], (const float*)c->d_sh_sdf[0], (const float*)c->d_sh_uw[0], c->d_sh_gkey[1],
                       c->d_sh_seq[1], c->d_sh_sdf[1], c->d_sh_uw[1]);
    HIPCHK(c, hipMemcpyAsync(c->sh_counts, c->d_sh_counts, 65 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  return KS_OK;
}

int frame_tail(ks_ctx* c, FrameSlot& S) {
  if (!S.pending) return KS_OK;
  S.pending = false;
  // After a pool / index failure the table may hold entries without a tile: frames that were
  // already in flight are dropped, never applied (the error has been reported for the frame that hit it).
  if (c->fatal) return KS_OK;
  if (!S.b_launched) {  // the frame's batch has not filled up (flush, or a lag shorter than the batch): it goes out as it is
    if (int rc = launch_batch(c)) return rc;
  }
  hipStream_t st = c->stream_tail;  // the host wait below orders the tail after the slot's front
  const FrameParams& F = S.F;
  {
    const auto w0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipEventSynchronize(S.ready));  // the frame's only host wait
    if (c->profiling) c->prof.host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
  HostTimer htt(&c->hp_t);
  Counters cnt = S.counters();
  uint32_t new_tiles = std::min(S.n_tiles(), c->cfg.max_tiles);
  const uint32_t tiles_before = c->tiles_initialised;
  const int set = S.prof_set;
  stage_mark(c, set, 6);
  if (c->eo_device && !c->eo_device_off) {
    c->eo_frames += 1;
    c->eo_iterations += S.h_snap->pad[2];   // rounds of the event-driven fix point (k_publish)
  }
  if (c->eo_device && !c->eo_device_off && !(cnt.err & kErrExact)) c->eo_hopeless.store(0, std::memory_order_relaxed);
  if (c->eo_trace && c->eo_device && !c->eo_device_off && !(cnt.err & kErrExact) && S.d_eo_ctl) {   // KS_EXACT_TRACE=1 (diagnostics; a host wait)
    EoCtl hctl;
    (void)hipStreamSynchronize(st);
    if (hipMemcpy(&hctl, S.d_eo_ctl, sizeof(hctl), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "[ks exact] frame %llu on the device: marks %llu / cap %zu, X %u / cap %zu, rounds %u; lists:", (unsigned long long)S.frame_no,
              (unsigned long long)hctl.st.n_marks, c->eo_cap_marks, hctl.n_x, c->eo_cap_x, hctl.rounds);
      for (int r = 1; r <= c->eo_bulk_rounds + 1 && r < (int)kEoBulkMax + 2; ++r) fprintf(stderr, " %u", hctl.n_in[r]);
      fprintf(stderr, "; sweeps (rays changed):");
      for (int i = 0; i < 32; ++i) fprintf(stderr, " %u%s", hctl.dense_chg[i] & 0x7fffffffu, (hctl.dense_chg[i] >> 31) ? "F" : "");
      fprintf(stderr, "\n");
    }
  }
  if ((cnt.err & kErrExact) && !(cnt.err & (kErrLabel | kErrIndex))) {
    // The device-driven fix point gave up (marks or X marks did not fit, the finisher ran out of rounds, or the frame
    // before this one fell back and had not entered its marks yet): the host-driven loop repeats the fix point from
    // whatever lengths the slot holds (any seed converges), enters the marks, and scan + emission run again — as after a
    // pair-buffer overflow, on the tail stream only.  Nothing was emitted and no tile was allocated by this frame.
    int rc;
    HIPCHK(c, hipStreamSynchronize(st));
    EoCtl hctl;
    HIPCHK(c, hipMemcpy(&hctl, S.d_eo_ctl, sizeof(hctl), hipMemcpyDeviceToHost));
    // (the largest request of the frames that failed since the buffers last grew: a later frame's smaller one must not replace it)
    auto want_at_least = [](std::atomic<size_t>& w, size_t v) {
      size_t cur = w.load(std::memory_order_relaxed);
      while (cur < v && !w.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
    };
    if (hctl.fail & kEoFailMarks)
      want_at_least(c->eo_want_marks, std::max<size_t>(2 * c->eo_cap_marks, (size_t)hctl.st.n_marks + (size_t)hctl.st.n_marks / 4));
    // X marks are for the few rays the seed stopped too early; a frame that wants more of them than an eighth of its
    // marks (2 cm voxels / 10 m rays: the approximate set is overwhelmed, the seed is wrong on most rays) is not a sparse
    // problem, and neither is one whose lists are still long after the bulk rounds
    const bool x_dense = (hctl.fail & kEoFailX) && (size_t)hctl.n_x > (size_t)hctl.st.n_marks / 8;
    // lists still long when the finisher takes over: more rounds as launches for the frames to come, while there is room
    const bool more_bulk = (hctl.fail & kEoFailRounds) && !x_dense && c->eo_bulk_rounds < (int)kEoBulkMax;
    if (more_bulk) {
      const int w = std::min((int)kEoBulkMax, c->eo_bulk_rounds + std::max(4, c->eo_bulk_rounds / 2));
      int cur = c->eo_want_bulk.load(std::memory_order_relaxed);
      while (cur < w && !c->eo_want_bulk.compare_exchange_weak(cur, w, std::memory_order_relaxed)) {}
    }
    const bool dense = x_dense || ((hctl.fail & kEoFailRounds) && !more_bulk);
    // (the count at the moment of the failure is a lower bound — the rounds stop there — and every growth costs the frames in
    // flight a repetition on the host: grow generously, a node is 16 bytes)
    if ((hctl.fail & kEoFailX) && !dense) want_at_least(c->eo_want_x, std::max<size_t>(16 * c->eo_cap_x, 8 * (size_t)hctl.n_x));
    if (dense) c->eo_hopeless.fetch_add(1, std::memory_order_relaxed);
    else if (!(hctl.fail & kEoFailChain)) c->eo_hopeless.store(0, std::memory_order_relaxed);
    c->eo_fallbacks.fetch_add(1, std::memory_order_relaxed);
    if (c->eo_trace) {   // KS_EXACT_TRACE=1 (diagnostics): why the device loop gave this frame up
      fprintf(stderr, "[ks exact] frame %llu falls back: fail %x (1 marks, 2 X marks, 4 rounds, 8 chain) marks %llu / cap %zu, X %u / cap %zu, rounds %u, dense %d; lists:",
              (unsigned long long)S.frame_no, hctl.fail, (unsigned long long)hctl.st.n_marks, c->eo_cap_marks, hctl.n_x, c->eo_cap_x, hctl.rounds, (int)dense);
      for (int r = 1; r <= c->eo_bulk_rounds + 1 && r < (int)kEoBulkMax + 2; ++r) fprintf(stderr, " %u", hctl.n_in[r]);
      fprintf(stderr, "; sweeps (rays changed):");
      for (int i = 0; i < 32; ++i) fprintf(stderr, " %u%s", hctl.dense_chg[i] & 0x7fffffffu, (hctl.dense_chg[i] >> 31) ? "F" : "");
      fprintf(stderr, "\n");
    }
    Counters rcnt{};
    rcnt.n_rays = cnt.n_rays;
    HIPCHK(c, hipMemcpyAsync(c->d_retry_counters, &rcnt, sizeof(rcnt), hipMemcpyHostToDevice, st));
    if ((rc = exact_early_out(c, S, st, c->d_retry_counters))) return rc;
    HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)c->d_eo_committed, (int)(S.F.eo_frame + 1u), 1, st));
    {
      BatchView V{};
      V.s[0] = slot_view(S, c->d_retry_counters);
      hipLaunchKernelGGL(k_scan_local, dim3((uint32_t)((c->cap_points + kScanBlock - 1) / kScanBlock), 1), dim3(1024), 0, st, V);
      launch_emit(c, V, 1, S.wide, st);
    }
    HIPCHK(c, hipMemcpyAsync(&rcnt, c->d_retry_counters, sizeof(rcnt), hipMemcpyDeviceToHost, st));
    uint32_t nt = 0;
    HIPCHK(c, hipMemcpyAsync(&nt, c->table.n_tiles, sizeof(nt), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    cnt.err = rcnt.err;
    cnt.n_pairs = rcnt.n_pairs;
    new_tiles = std::min(nt, c->cfg.max_tiles);
  }
  c->pairs_hint.store(std::max<size_t>(c->pairs_hint.load(std::memory_order_relaxed), cnt.n_pairs), std::memory_order_relaxed);
  if ((cnt.err & kErrPairs) && !(cnt.err & ~kErrPairs)) {
    // The frame's pairs did not fit the buffer sized from earlier frames: nothing was written and no tile
    // was allocated.  Grow it and repeat the emission (the scan of the counts is still in the slot).
    // Later frames may already have allocated tiles; the emission is a get-or-insert, so that is harmless.
    // (This runs on the helper thread while the caller may be capturing stage B of another slot on a march
    // stream: nothing here may touch a march stream.  None has to: S.ready has ordered this slot's stage B, the
    // tail stream is this thread's own, and no other frame reads this slot's pair buffer.)
    int rc;
    HIPCHK(c, hipStreamSynchronize(st));
    if ((rc = ensure_pairs_in(c, S, (size_t)cnt.n_pairs + (size_t)cnt.n_pairs / 4))) return rc;
    Counters rcnt{};
    rcnt.n_rays = cnt.n_rays;
    HIPCHK(c, hipMemcpyAsync(c->d_retry_counters, &rcnt, sizeof(rcnt), hipMemcpyHostToDevice, st));
    {
      BatchView V{};
      V.s[0] = slot_view(S, c->d_retry_counters);
      launch_emit(c, V, 1, S.wide, st);
    }
    HIPCHK(c, hipMemcpyAsync(&rcnt, c->d_retry_counters, sizeof(rcnt), hipMemcpyDeviceToHost, st));
    uint32_t nt = 0;
    HIPCHK(c, hipMemcpyAsync(&nt, c->table.n_tiles, sizeof(nt), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    cnt.err = rcnt.err;
    cnt.n_pairs = rcnt.n_pairs;
    new_tiles = std::min(nt, c->cfg.max_tiles);
  }
  // tiles allocated by the front exist in the table whatever happens next: make them valid
  if (new_tiles > c->tiles_initialised) {
    hipLaunchKernelGGL(k_init_tiles, dim3(new_tiles - c->tiles_initialised), dim3(512), 0, st, c->pool,
                       c->tiles_initialised);
    c->tiles_initialised = new_tiles;
  }
  auto finish_prof = [&](uint64_t n_pairs) {
    if (set < 0) return;
    ProfSet& P = c->pset[set];
    P.n_pairs = n_pairs;
    (void)hipEventRecord(P.ev[kStageEvents - 1], st);
    P.complete = true;
  };
  if (cnt.err & kErrLabel) {
    for (int e = 7; e < kStageEvents - 1; ++e) stage_mark(c, set, e);
    finish_prof(0);
    c->err = "semantic label >= 21 (CHECK_LT in the reference)";
    return KS_ERR_LABEL_RANGE;
  }
  if (cnt.err) {
    for (int e = 7; e < kStageEvents - 1; ++e) stage_mark(c, set, e);
    finish_prof(0);
    c->fatal = true;
    if (cnt.err & kErrPool) {
      c->err = "voxel tile pool exhausted: raise ks_config.max_tiles";
      return KS_ERR_POOL_FULL;
    }
    c->err = "voxel index out of the packed range / tile table full";
    return KS_ERR_INDEX_RANGE;
  }
  const unsigned long long n_pairs = cnt.n_pairs;
  if (c->shard_export) {
    // a marcher of ks_integrate_round_exact: the frame's updates leave as records, grouped by the rank that owns their tile
    for (int e = 7; e < kStageEvents - 1; ++e) stage_mark(c, set, e);
    if (int rc = shard_export_frame(c, S, n_pairs, st)) return rc;
    finish_prof(n_pairs);
    HIPCHK(c, hipEventRecord(S.tail_done, st));
    S.tail_recorded = true;
    c->owed.n_points += S.n;
    c->owed.n_valid_points += cnt.n_valid;
    c->owed.n_rays_cast += cnt.n_rays;
    c->owed.n_voxel_updates += n_pairs;
    return KS_OK;
  }
  if (n_pairs == 0 && c->pending_join) {
    // a frame without updates still separates the frame before it from the one after it, which share a parity
    // buffer set: the long runs of the previous frame end before anything later is enqueued on the tail stream
    HIPCHK(c, hipStreamWaitEvent(st, c->pending_join, 0));
    c->pending_join = nullptr;
  }
  if (n_pairs > 0) {
    int rc;
    if ((rc = ensure_pairs_out(c, n_pairs))) return rc;
    stage_mark(c, set, 7);
    const unsigned end_bit = F.seq_bits + 9 + bits_for(new_tiles);
    uint64_t* sp = nullptr;
    // k_emit wrote the pairs in integration order; the stable sort only groups them by voxel (it skips
    // the sequence bits), so every voxel replays its updates in the reference's single-thread order.
    const int par = (int)(S.frame_no & 1u);
    uint64_t* const d_pairs2 = c->d_pairs2_[par];
    unsigned long long* const d_long_list = c->d_long_list_[par];
    // Pipelined contexts without an early-out (stage B follows stage A on the front stream): the pair sort — bound by HBM — goes to
    // the FRONT stream, the update — bound by resident workgroups — stays on the tail stream, so that the sort of frame f runs beside
    // the update of frame f - 1 (on one stream they ran one after the other: the tail stream was the frame period).  The sort waits for
    // the update that last read the buffer set it writes (frame f - 2: the sets alternate), the tail stream for the sort.
    const bool sort_front = c->sort_on_front && st != c->stream && !c->uses_early_out && !(set >= 0 && c->pset[set].stages);
    if (sort_front) {
      hipStream_t ss = c->stream;
      FrameSlot* prev = c->last_tail_of_parity[par];
      if (prev && prev != &S) {
        if (prev->tail_recorded) HIPCHK(c, hipStreamWaitEvent(ss, prev->tail_done, 0));
        if (prev->join_recorded) HIPCHK(c, hipStreamWaitEvent(ss, prev->join, 0));
      }
      HostTimer ht(&c->hp_sort);
      HIPCHK(c, (ksrs::sort<uint64_t, false>(c->sort_ws_tail, S.d_pairs, d_pairs2, nullptr, nullptr, (size_t)n_pairs, std::min(56u, end_bit), ss, &sp,
                                             nullptr, F.seq_bits)));
      HIPCHK(c, hipEventRecord(S.sorted, ss));
      HIPCHK(c, hipStreamWaitEvent(st, S.sorted, 0));
    } else if ((rc = sort_keys(c, S.d_pairs, d_pairs2, n_pairs, std::min(56u, end_bit), &sp, F.seq_bits, /*tail=*/true))) {
      return rc;
    }
    c->last_tail_of_parity[par] = &S;
    stage_mark(c, set, 8);
    const uint32_t ab = (uint32_t)((n_pairs + 255) / 256);
    const uint32_t run_tile = kRunPer * c->run_threads;
    const uint32_t rb = (uint32_t)((n_pairs + run_tile - 1) / run_tile);
    const bool by_runs = n_pairs >= c->apply_runs_min_pairs;
    const uint32_t lb = (uint32_t)std::min<unsigned long long>(n_pairs / kLongRun + 1, 4096);
    const bool time_apply = set >= 0 && c->pset[set].apply;
    if (time_apply) c->pset[set].applied = true;
    // long runs (voxels next to the sensor) are listed first; then the two update kernels run side by side:
    // k_apply on the tail stream, k_apply_long on its own stream (disjoint voxels)
    hipStream_t sl = c->stream_long ? c->stream_long : st;
    hipStream_t sx = (sl != st && c->stream_xlong) ? c->stream_xlong : nullptr;
    unsigned long long* const d_xlong_list = sx ? d_long_list + (c->cap_pairs / kLongRunLanes + 64) : nullptr;
    const bool lanes_on = by_runs && sl != st && sx && c->long_lanes && n_pairs >= c->long_lanes_min_pairs;
    // ("long" could begin at kLongRunLanes = 17 updates where the lanes kernel takes the long runs — k_find_long, k_long_measure and
    // k_apply_runs take the threshold as an argument — but measured at 1280x720 / 2 cm it buys nothing: k_apply_runs 2.83 vs 2.79 ms,
    // the lanes kernel 1.47 vs 0.93 ms, the frame 6.42 vs 6.24 ms: profiles/r06_c4_merged_ab.txt.  KS_LONG_MIN=16 selects it.)
    const uint32_t long_min = (lanes_on && c->long_min_lanes) ? kLongRunLanes : kLongRun;
    // k_apply_runs decides which runs are its own by itself: the listing of the long runs (a pass over all pairs, 0.3 ms at
    // 1280x720 / 2 cm) then runs BESIDE it, on the long-run stream, instead of in front of it
    const bool find_beside = by_runs && sl != st && sx;
    const dim3 find_grid((uint32_t)((n_pairs + 256 * kFindLongItems - 1) / (256 * kFindLongItems)));
    if (!find_beside)
      hipLaunchKernelGGL(k_find_long, find_grid, dim3(256), 0, st, F.seq_bits, n_pairs, (const uint64_t*)sp, d_long_list, d_xlong_list,
                         S.d_counters, long_min);
    const uint32_t xb = (uint32_t)std::min<unsigned long long>(n_pairs / kXLongRun + 1, 512);
    if (sl != st) {
      // the previous frame's long runs end before any voxel of this frame is touched
      if (c->pending_join) HIPCHK(c, hipStreamWaitEvent(st, c->pending_join, 0));
      c->pending_join = nullptr;
      HIPCHK(c, hipEventRecord(S.fork, st));
      HIPCHK(c, hipStreamWaitEvent(sl, S.fork, 0));
      if (lanes_on) HIPCHK(c, hipMemsetAsync(c->d_long_hdr_[par], 0, sizeof(LongHdr), sl));
      if (find_beside) {
        hipLaunchKernelGGL(k_find_long, find_grid, dim3(256), 0, sl, F.seq_bits, n_pairs, (const uint64_t*)sp, d_long_list, d_xlong_list,
                           S.d_counters, long_min);
        HIPCHK(c, hipEventRecord(S.found, sl));
        HIPCHK(c, hipStreamWaitEvent(sx, S.found, 0));
      } else if (sx) {
        HIPCHK(c, hipStreamWaitEvent(sx, S.fork, 0));
      }
    }
    // k_apply_runs on a stream of its own (pipelined contexts with the long-run streams; not while the stages are being timed):
    // the sort is bound by HBM, this kernel by resident workgroups — the next frame's sort runs beside it instead of behind it.
    // It is ordered like the long-run kernels: after this frame's fork, before S.join (which the next frame's fork waits for).
    const bool apply_beside = find_beside && c->stream_apply && c->defer_join && !(set >= 0 && c->pset[set].stages);
    hipStream_t sa = apply_beside ? c->stream_apply : st;
    if (apply_beside) HIPCHK(c, hipStreamWaitEvent(sa, S.fork, 0));
#define KS_LAUNCH_RUNS(MODE, MERGED, TH)                                                                              \
  if (time_apply)                                                                                                     \
    hipExtLaunchKernelGGL((k_apply_runs<MODE, MERGED, TH>), dim3(rb), dim3(TH), 0, sa, c->pset[set].k0, c->pset[set].k1, \
                          0, F, n_pairs, sp, S.d_rays, S.d_deltas, c->table, c->pool, c->d_label_lut, long_min);       \
  else                                                                                                                \
    hipLaunchKernelGGL((k_apply_runs<MODE, MERGED, TH>), dim3(rb), dim3(TH), 0, sa, F, n_pairs, sp, S.d_rays,          \
                       S.d_deltas, c->table, c->pool, c->d_label_lut, long_min)
#define KS_LAUNCH_APPLY_M(MODE, MERGED)                                                                              \
  if (by_runs && c->run_threads == 256u) {                                                                           \
    KS_LAUNCH_RUNS(MODE, MERGED, 256u);                                                                              \
  } else if (by_runs) {                                                                                              \
    KS_LAUNCH_RUNS(MODE, MERGED, 512u);                                                                              \
  }                                                                                                                  \
  else if (time_apply)                                                                                               \
    hipExtLaunchKernelGGL((k_apply<MODE, MERGED>), dim3(ab), dim3(256), 0, st, c->pset[set].k0, c->pset[set].k1, 0,   \
                          F, n_pairs, sp, S.d_rays, S.d_deltas, c->table, c->pool, c->d_label_lut, d_long_list,       \
                          S.d_counters);                                                                              \
  else                                                                                                               \
    hipLaunchKernelGGL((k_apply<MODE, MERGED>), dim3(ab), dim3(256), 0, st, F, n_pairs, sp, S.d_rays, S.d_deltas,     \
                       c->table, c->pool, c->d_label_lut, d_long_list, S.d_counters)
#define KS_LAUNCH_APPLY(MODE)                                                                                        \
  if (c->cfg.method == KS_METHOD_MERGED) {                                                                           \
    KS_LAUNCH_APPLY_M(MODE, true);                                                                                   \
  } else {                                                                                                           \
    KS_LAUNCH_APPLY_M(MODE, false);                                                                                  \
  }                                                                                                                  \
  stage_mark(c, set, 9);                                                                                             \
  if (sx && c->xl_parallel && n_pairs >= c->xl_min_pairs) {                                                          \
    /* the class sums and the weight of such runs as integer sums per chunk, chunks side by side (ks_k_apply_xl.h); what \
       the shortcut cannot carry goes to k_apply_xlong through the fall-back list */                                    \
    hipLaunchKernelGGL(k_xl_measure<MODE>, dim3(kXlMaxRuns / 256), dim3(256), 0, sx, F, n_pairs, (const uint64_t*)sp,  \
                       c->pool, (const unsigned long long*)d_xlong_list, (const Counters*)S.d_counters, c->d_xl_runs); \
    hipLaunchKernelGGL(k_xl_number, dim3(1), dim3(1024), 0, sx, (const unsigned long long*)d_xlong_list,               \
                       (const Counters*)S.d_counters, c->d_xl_runs, c->d_xl_idx, c->d_xl_fb, c->d_xl_hdr,              \
                       c->cap_xl_chunks);                                                                               \
    hipLaunchKernelGGL(k_xl_chunks, dim3(4096), dim3(256), 0, sx, F, (const uint64_t*)sp, (const RayDesc*)S.d_rays,    \
                       (const float*)S.d_deltas, c->table, c->d_xl_runs, (const uint32_t*)c->d_xl_idx,                 \
                       (const XlHeader*)c->d_xl_hdr, c->d_xl_chunks);                                                   \
    hipLaunchKernelGGL(k_xl_walk<MODE>, dim3(2048), dim3(64), 0, sx, F, (const uint64_t*)sp, (const RayDesc*)S.d_rays, \
                       (const float*)S.d_deltas, c->table, c->pool, (const uint32_t*)c->d_label_lut, c->d_xl_runs,     \
                       (const uint32_t*)c->d_xl_idx, c->d_xl_hdr, (const XlChunk*)c->d_xl_chunks, c->d_xl_fb);         \
    hipLaunchKernelGGL(k_apply_xlong<MODE>, dim3(xb), dim3(256), 0, sx, F, n_pairs, sp, S.d_rays, S.d_deltas,          \
                       c->table, c->pool, c->d_label_lut, (const unsigned long long*)c->d_xl_fb,                        \
                       (const uint32_t*)&c->d_xl_hdr->n_fallback);                                                      \
  } else if (sx)                                                                                                     \
    hipLaunchKernelGGL(k_apply_xlong<MODE>, dim3(xb), dim3(256), 0, sx, F, n_pairs, sp, S.d_rays, S.d_deltas,            \
                       c->table, c->pool, c->d_label_lut, d_xlong_list, (const uint32_t*)&S.d_counters->n_xlong);      \
  if (lanes_on) {                                                                                                    \
    const uint32_t cap_long = (uint32_t)(n_pairs / (long_min + 1) + 1);                                               \
    hipLaunchKernelGGL(k_long_measure, dim3((cap_long + 255) / 256), dim3(256), 0, sl, F.seq_bits, n_pairs,           \
                       (const uint64_t*)sp, d_long_list, (const Counters*)S.d_counters, c->d_long_hdr_[par], long_min); \
    hipLaunchKernelGGL(k_long_bucket, dim3((cap_long + 255) / 256), dim3(256), 0, sl,                                 \
                       (const unsigned long long*)d_long_list, (const Counters*)S.d_counters, c->d_long_hdr_[par],    \
                       c->d_long_sorted_[par]);                                                                        \
    if (c->lanes_depth == 4u)                                                                                        \
      hipLaunchKernelGGL((k_apply_long_lanes<MODE, 4u>), dim3((cap_long / 64 + kLongClasses + 3) / 4), dim3(256), 0, sl, F, \
                         (const uint64_t*)sp, (const RayDesc*)S.d_rays, (const float*)S.d_deltas, c->table, c->pool,   \
                         (const uint32_t*)c->d_label_lut, (const LongHdr*)c->d_long_hdr_[par],                         \
                         (const unsigned long long*)c->d_long_sorted_[par]);                                           \
    else                                                                                                             \
      hipLaunchKernelGGL((k_apply_long_lanes<MODE, 6u>), dim3((cap_long / 64 + kLongClasses + 3) / 4), dim3(256), 0, sl, F, \
                         (const uint64_t*)sp, (const RayDesc*)S.d_rays, (const float*)S.d_deltas, c->table, c->pool,   \
                         (const uint32_t*)c->d_label_lut, (const LongHdr*)c->d_long_hdr_[par],                         \
                         (const unsigned long long*)c->d_long_sorted_[par]);                                           \
    hipLaunchKernelGGL(k_apply_long<MODE>, dim3(std::min<uint32_t>(lb, 1024u)), dim3(128), 0, sl, F, n_pairs, sp,     \
                       S.d_rays, S.d_deltas, c->table, c->pool, c->d_label_lut,                                        \
                       (const unsigned long long*)c->d_long_sorted_[par], (const Counters*)S.d_counters,               \
                       (const LongHdr*)c->d_long_hdr_[par]);                                                           \
  } else                                                                                                             \
  hipLaunchKernelGGL(k_apply_long<MODE>, dim3(lb), dim3(128), 0, sl, F, n_pairs, sp, S.d_rays, S.d_deltas,               \
                     c->table, c->pool, c->d_label_lut, d_long_list, S.d_counters)
    switch (c->cfg.color_mode) {
      case KS_COLOR_MODE_COLOR: KS_LAUNCH_APPLY(KS_COLOR_MODE_COLOR); break;
      case KS_COLOR_MODE_SEMANTIC: KS_LAUNCH_APPLY(KS_COLOR_MODE_SEMANTIC); break;
      default: KS_LAUNCH_APPLY(KS_COLOR_MODE_SEMANTIC_PROBABILITY); break;
    }
#undef KS_LAUNCH_APPLY
#undef KS_LAUNCH_APPLY_M
#undef KS_LAUNCH_RUNS
    if (sl != st) {
      if (sx) {  // S.join stands for both lists
        HIPCHK(c, hipEventRecord(S.join_x, sx));
        HIPCHK(c, hipStreamWaitEvent(sl, S.join_x, 0));
      }
      if (apply_beside) {  // ... and for k_apply_runs
        HIPCHK(c, hipEventRecord(S.applied, sa));
        HIPCHK(c, hipStreamWaitEvent(sl, S.applied, 0));
      }
      HIPCHK(c, hipEventRecord(S.join, sl));
      S.join_recorded = true;
      // deferred: the tail stream goes on with the next frame's tile initialisation, pair sort and long-run
      // listing (none of which touches voxels or this frame's buffer set) and waits before its k_apply
      if (c->defer_join && !(set >= 0 && c->pset[set].stages)) c->pending_join = S.join;
      else HIPCHK(c, hipStreamWaitEvent(st, S.join, 0));
    }
  } else {
    stage_mark(c, set, 7);
    stage_mark(c, set, 8);
    stage_mark(c, set, 9);
  }
  finish_prof(n_pairs);
  HIPCHK(c, hipEventRecord(S.tail_done, st));
  S.tail_recorded = true;
  HIPCHK(c, hipGetLastError());
  c->owed.n_points += S.n;
  c->owed.n_valid_points += cnt.n_valid;
  c->owed.n_rays_cast += cnt.n_rays;
  c->owed.n_voxel_updates += n_pairs;
  c->owed.n_blocks_allocated += new_tiles > tiles_before ? new_tiles - tiles_before : 0u;  // (marches of later frames run ahead)
  if (c->use_bundle_rank && !c->bo_hint_fixed) {
    // the bundle count the next frames' epochs are launched for: this frame's rays (= bundles of both maps) + 25 % + 2048, decaying slowly
    const uint32_t want = cnt.n_rays + cnt.n_rays / 4u + 2048u;
    const uint32_t cur = c->bo_hint.load(std::memory_order_relaxed);
    c->bo_hint.store(cur == ~0u ? want : std::max(want, cur - cur / 64u), std::memory_order_relaxed);
  }
  return KS_OK;
}

// hand the statistics of every frame completed since the last hand-over to the caller
void deliver_stats(ks_ctx* c, ks_frame_stats* stats) {
  if (stats) *stats = c->owed;
  c->owed = ks_frame_stats{};
}

// run the tail of a frame whose front is still waiting for it (pipelined mode)
int flush_pending(ks_ctx* c) {
  // up to two slots are pending between calls; oldest frame first
  for (int k = 0; k < c->n_slots; ++k) {
    FrameSlot& S = c->slot[(c->frame_no + k) % (uint64_t)c->n_slots];
    if (S.pending) {
      const int rc = frame_tail(c, S);
      if (rc) return rc;
    }
  }
  return KS_OK;
}

int ensure_exchange(ks_ctx* c, size_t n) {
  if (n <= c->cap_xchg) return KS_OK;
  const size_t cap = std::max<size_t>(n + n / 2, 1024);
  int rc;
  if ((rc = dev_alloc(c, &c->d_xchg_u32, 2 * cap + 2))) return rc;
  if ((rc = dev_alloc(c, &c->d_xchg_u64, cap))) return rc;
  c->cap_xchg = cap;
  return KS_OK;
}

// complete every outstanding frame and drain both streams
int quiesce(ks_ctx* c) {
  const int rc = flush_pending(c);
  if (c->stream_tail != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_tail));
  if (c->stream_long) HIPCHK(c, hipStreamSynchronize(c->stream_long));
  c->pending_join = nullptr;
  if (int rc2 = sync_march(c)) return rc2;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return rc;
}

void tail_worker(ks_ctx* c) {
  (void)hipSetDevice(c->cfg.device_id);
  std::unique_lock<std::mutex> lk(c->tail_mu);
  for (;;) {
    c->tail_cv.wait(lk, [&] { return c->tail_job != nullptr || c->tail_quit; });
    if (c->tail_quit) return;
    FrameSlot* S = c->tail_job;
    c->tail_job = nullptr;
    lk.unlock();
    int rc;
    {
      std::lock_guard<std::mutex> cap(c->capture_mu);
      rc = frame_tail(c, *S);
    }
    lk.lock();
    c->tail_rc = rc;
    c->tail_busy = false;
    c->tail_cv.notify_all();
  }
}
void tail_post(ks_ctx* c, FrameSlot* S) {
  std::lock_guard<std::mutex> lk(c->tail_mu);
  c->tail_job = S;
  c->tail_busy = true;
  c->tail_rc = KS_OK;
  c->tail_cv.notify_all();
}
int tail_join(ks_ctx* c) {
  std::unique_lock<std::mutex> lk(c->tail_mu);
  c->tail_cv.wait(lk, [&] { return !c->tail_busy; });
  return c->tail_rc;
}

// The reference allocates blocks on demand without a cap [K:src/semantic_integrator_base.cpp:205-254]; the tile
// pool here is one allocation.  When more than half of it is in use it is doubled between frames (new
// allocation, device-to-device copy of the tiles in use, table rebuilt with the same slot numbers), so a map
// only stops growing when HBM is exhausted.  ks_config.max_tiles is the INITIAL capacity.
int grow_pool(ks_ctx* c) {
  int rc;
  if ((rc = quiesce(c))) return rc;
  const size_t old_max = c->cfg.max_tiles;
  const size_t new_max = std::min<size_t>(old_max * 2, (1u << 23) - 1);
  if (new_max <= old_max) return KS_OK;
  const uint32_t nt = c->tiles_initialised;
  uint4* vox = nullptr;
  uint8_t *upd = nullptr, *dirty = nullptr;
  uint64_t* skeys = nullptr;
  TileEntry* ent = nullptr;
  uint32_t cap = 1024;
  while (cap < 2u * new_max) cap <<= 1;
  if (hipMalloc((void**)&vox, new_max * kTileVoxels * 8 * sizeof(uint4)) != hipSuccess) {
    (void)hipGetLastError();
    return KS_OK;  // no memory for a bigger pool: carry on with the current one (exhaustion is reported when it happens)
  }
  // (a failure part-way leaves the old pool in place and frees what was allocated for the new one)
  auto fill = [&]() -> hipError_t {
    hipError_t e;
    if ((e = hipMalloc((void**)&upd, new_max)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&dirty, new_max)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&skeys, new_max * sizeof(uint64_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&ent, (size_t)cap * sizeof(TileEntry))) != hipSuccess) return e;
    if ((e = hipMemset(upd, 0, new_max)) != hipSuccess) return e;
    if ((e = hipMemset(dirty, 0, new_max)) != hipSuccess) return e;
    if ((e = hipMemset(ent, 0xff, (size_t)cap * sizeof(TileEntry))) != hipSuccess) return e;
    if (nt) {
      if ((e = hipMemcpy(vox, c->pool.vox, (size_t)nt * kTileVoxels * 8 * sizeof(uint4), hipMemcpyDeviceToDevice)) != hipSuccess) return e;
      if ((e = hipMemcpy(upd, c->pool.updated, nt, hipMemcpyDeviceToDevice)) != hipSuccess) return e;
      if ((e = hipMemcpy(dirty, c->pool.dirty, nt, hipMemcpyDeviceToDevice)) != hipSuccess) return e;
      if ((e = hipMemcpy(skeys, c->table.slot_keys, (size_t)nt * sizeof(uint64_t), hipMemcpyDeviceToDevice)) != hipSuccess) return e;
    }
    return hipSuccess;
  };
  if (const hipError_t e = fill(); e != hipSuccess) {
    for (void* q : {(void*)vox, (void*)upd, (void*)dirty, (void*)skeys, (void*)ent})
      if (q) (void)hipFree(q);
    c->err = std::string("grow_pool: ") + hipGetErrorString(e);
    return KS_ERR_HIP;
  }
  (void)hipFree(c->pool.vox);
  (void)hipFree(c->pool.updated);
  (void)hipFree(c->pool.dirty);
  (void)hipFree(c->table.slot_keys);
  (void)hipFree(c->table.ent);
  c->pool.vox = vox;
  c->pool.updated = upd;
  c->pool.dirty = dirty;
  c->table.slot_keys = skeys;
  c->table.ent = ent;
  c->table.mask = cap - 1;
  c->table.max_tiles = (uint32_t)new_max;
  c->cfg.max_tiles = (uint32_t)new_max;
  if (nt) hipLaunchKernelGGL(k_rehash_tiles, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->table, (const uint64_t*)skeys, nt);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ++c->buffers_epoch;  // captured graphs hold the old table / pool
  return KS_OK;
}

int integrate_device_impl(ks_ctx* c, const float Tq[7], const float* d_xyz, const uint8_t* d_rgba, const uint8_t* d_labels,
                          size_t n, int freespace, ks_frame_stats* stats) {
  if (c->fatal) {
    c->err = "context is in a failed state (earlier pool/index error)";
    return KS_ERR_INVALID_ARG;
  }
  if (n >= (1u << 23)) {
    c->err = "more than 2^23-1 points per call";
    return KS_ERR_INVALID_ARG;
  }
  if (c->uses_early_out && n > kObsMaxPoints) {
    c->err = "fast integrator with the early-out enabled: at most 2^22-2 points per call";
    return KS_ERR_UNSUPPORTED;
  }
  const ks_config& cfg = c->cfg;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  int rc;
  if (c->eo_device && !c->eo_device_off && c->eo_hopeless.load(std::memory_order_relaxed) >= 3) {
    // the device loop keeps giving up on this context's frames: the host-driven loop from here on, one frame at a time
    if ((rc = quiesce(c))) return rc;
    c->eo_device_off = true;
    c->cfg.pipeline_frames = 0;
  }
  // sorted integration order keeps its permutation in single buffers: not pipelined
  bool pipelined = cfg.pipeline_frames && cfg.integration_order_mode != KS_ORDER_SORTED;
  if (pipelined && c->eo_device && !c->eo_device_off) {
    // A frame that fell back enters its marks when its tail runs, `pipeline_frames` calls after its stage B — so the frames
    // in flight behind it find their predecessor's marks missing when their finisher runs and follow it to the host-driven
    // loop, and so would every frame after them, for ever.  Complete what is in flight once and start afresh.
    const uint64_t fb = c->eo_fallbacks.load(std::memory_order_relaxed);
    if (fb != c->eo_fallbacks_seen) {
      if ((rc = quiesce(c))) return rc;
      c->eo_fallbacks_seen = c->eo_fallbacks.load(std::memory_order_relaxed);
    }
  }

  // frame-level bookkeeping of the fast integrator [K:src/semantic_tsdf_integrator_fast.cpp:165-170]
  if (cfg.method == KS_METHOD_FAST) {
    if ((++c->reset_counter) >= cfg.clear_checks_every_n_frames) {
      c->reset_counter = 0;
      if ((rc = reset_set(c, c->d_start_set, &c->start_offset, false))) return rc;
      if ((rc = reset_set(c, nullptr, &c->observed_offset, true))) return rc;
    }
    ++c->obs_tag;  // this frame's marks
  }
  if (n == 0) {
    if ((rc = quiesce(c))) return rc;
    deliver_stats(c, stats);
    return KS_OK;
  }
  {
    // freshest tile count the host has seen (snapshots of frames whose tail is still to come included)
    uint32_t known = c->tiles_initialised;
    for (const FrameSlot& S2 : c->slot)
      if (S2.h_snap) known = std::max(known, S2.h_snap->n_tiles);
    if ((size_t)known * 2 > (size_t)c->cfg.max_tiles && (rc = grow_pool(c))) return rc;
  }
  if (n > c->cap_points) {  // growing frees buffers a pending tail still needs
    if ((rc = quiesce(c))) return rc;
    if ((rc = ensure_points(c, n))) return rc;
  }
  if (c->eo_device && !c->eo_device_off) {
    if (c->eo_want_marks.load(std::memory_order_relaxed) > c->eo_cap_marks || c->eo_want_x.load(std::memory_order_relaxed) > c->eo_cap_x) {
      if ((rc = quiesce(c))) return rc;   // (the frames in flight may ask for more while they are completed)
      const size_t wm = std::max(c->eo_want_marks.load(std::memory_order_relaxed), c->eo_cap_marks);
      // (a frame that wants more X marks than an eighth of its marks is not helped by room for them — frame_tail counts it
      // as hopeless — so the X buffers never need more than that)
      const size_t wx = std::max(std::min(c->eo_want_x.load(std::memory_order_relaxed), std::max<size_t>(wm / 8, (size_t)1 << 17)), c->eo_cap_x);
      if (ensure_exact_slots(c, wm, wx) != KS_OK) {
        // no memory for it: this context stays with the host-driven loop (one frame at a time, the buffers of ensure_marks)
        // instead of failing the frame.  The slots' buffers may be gone: nothing of the event-driven path is touched again.
        (void)hipGetLastError();
        c->eo_device_off = true;
        c->cfg.pipeline_frames = 0;
        c->batch = 1;          // (one frame at a time from here on — this frame included: nothing is in flight after the quiesce above)
        pipelined = false;
        c->err.clear();
      }
      c->eo_want_x.store(0, std::memory_order_relaxed);   // (what was asked for has been looked at: a clamped request must not come back every frame)
      c->eo_fallbacks_seen = c->eo_fallbacks.load(std::memory_order_relaxed);
    }
    if (const int wb = c->eo_want_bulk.load(std::memory_order_relaxed); wb > c->eo_bulk_rounds) {
      if ((rc = quiesce(c))) return rc;   // (the helper thread reads eo_bulk_rounds while it runs a tail)
      c->eo_bulk_rounds = wb;
      ++c->buffers_epoch;                 // the captured launch sequences hold the old number of rounds
    }
  }
  if (!pipelined) {
    if ((rc = quiesce(c))) return rc;
    FrameSlot& S = c->slot[0];
    if ((rc = frame_front(c, S, Tq, d_xyz, d_rgba, d_labels, n, freespace))) return rc;
    rc = frame_tail(c, S);
    deliver_stats(c, stats);
    return rc;
  }
  // pipelined: stages A and B of this frame first, then stage T of the frame `lag` calls back
  // (its snapshot is long there: the host never waits for the march that is still running, and the
  // next call can enqueue stage A while this frame's march is in flight); the statistics returned
  // are those of the frames completed here
  const uint64_t lag = (uint64_t)std::min(std::max(cfg.pipeline_frames, 1), kMaxLag);
  FrameSlot& S = c->slot[c->frame_no % (uint64_t)c->n_slots];
  if (S.pending && (rc = frame_tail(c, S))) return rc;  // cannot happen: the slot's frame is 4 calls old
  const uint64_t this_frame = c->frame_no;
  FrameSlot* due = (this_frame >= lag) ? &c->slot[(this_frame - lag) % (uint64_t)c->n_slots] : nullptr;
  if (due && !due->pending) due = nullptr;
  if (due && !due->b_launched && (rc = launch_batch(c))) return rc;  // (only with a lag shorter than the batch)
  if (due && c->use_tail_thread) {
    tail_post(c, due);  // the helper thread enqueues the tail of the frame `lag` calls back ...
    const int rc_front = frame_front(c, S, Tq, d_xyz, d_rgba, d_labels, n, freespace);  // ... while this one enqueues A and B
    rc = tail_join(c);
    if (rc_front) rc = rc_front;
  } else {
    if ((rc = frame_front(c, S, Tq, d_xyz, d_rgba, d_labels, n, freespace))) return rc;
    if (due) rc = frame_tail(c, *due);
  }
  deliver_stats(c, stats);
  return rc;
}

int integrate_device(ks_ctx* c, const float Tq[7], const float* d_xyz, const uint8_t* d_rgba, const uint8_t* d_labels,
                     size_t n, int freespace, ks_frame_stats* stats) {
  if (!c->profiling) return integrate_device_impl(c, Tq, d_xyz, d_rgba, d_labels, n, freespace, stats);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = integrate_device_impl(c, Tq, d_xyz, d_rgba, d_labels, n, freespace, stats);
  c->prof.host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int collect_block_indices(ks_ctx* c, bool only_updated, bool reset, std::vector<int32_t>* out) {
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint32_t nt = c->tiles_initialised;
  std::vector<uint64_t> keys(nt);
  std::vector<uint8_t> upd(nt);
  if (nt) {
    HIPCHK(c, hipMemcpy(keys.data(), c->table.slot_keys, nt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(upd.data(), c->pool.updated, nt, hipMemcpyDeviceToHost));
    if (only_updated && reset) HIPCHK(c, hipMemset(c->pool.updated, 0, nt));
  }
  std::set<std::tuple<int32_t, int32_t, int32_t>> s;
  for (uint32_t i = 0; i < nt; ++i) {
    if (only_updated && !upd[i]) continue;
    const uint64_t k = keys[i];
    const int tx = (int)((k >> 36) & 0x3ffffu) - kTileBias, ty = (int)((k >> 18) & 0x3ffffu) - kTileBias,
              tz = (int)(k & 0x3ffffu) - kTileBias;
    s.insert({tx >> c->vps_shift, ty >> c->vps_shift, tz >> c->vps_shift});
  }
  out->clear();
  for (const auto& t : s) {
    out->push_back(std::get<0>(t));
    out->push_back(std::get<1>(t));
    out->push_back(std::get<2>(t));
  }
  return KS_OK;
}

}  // namespace

// find-or-insert n tile keys (device array) and initialise the newly allocated tiles
static int insert_tiles(ks_ctx* c, const uint64_t* d_keys, size_t n, size_t distinct_at_most = ~(size_t)0) {
  int rc;
  if ((rc = quiesce(c))) return rc;
  // as for frames: keep at least half of the pool free for what is coming (all n keys may be new tiles — unless the caller knows
  // how many DISTINCT keys there can be: the records of a frame name a few thousand tiles a hundred times each)
  const size_t may_be_new = std::min(n, distinct_at_most);
  while ((size_t)c->tiles_initialised + may_be_new > (size_t)c->cfg.max_tiles / 2) {
    const uint32_t before = c->cfg.max_tiles;
    if ((rc = grow_pool(c))) return rc;
    if (c->cfg.max_tiles == before) break;  // at the limit, or no memory: exhaustion is reported if it happens
  }
  FrameSlot& S = c->slot[0];
  HIPCHK(c, hipMemsetAsync(S.d_counters, 0, sizeof(Counters), c->stream));
  hipLaunchKernelGGL(k_insert_tiles, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->table, S.d_counters,
                     d_keys, (uint32_t)n);
  {
    BatchView V{};
    V.s[0] = slot_view(S);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, V, (const uint32_t*)c->table.n_tiles);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint32_t new_tiles = std::min(S.n_tiles(), c->cfg.max_tiles);
  if (new_tiles > c->tiles_initialised) {
    hipLaunchKernelGGL(k_init_tiles, dim3(new_tiles - c->tiles_initialised), dim3(512), 0, c->stream, c->pool,
                       c->tiles_initialised);
    c->tiles_initialised = new_tiles;
  }
  if (S.counters().err) {
    c->fatal = true;
    c->err = "voxel tile pool exhausted: raise ks_config.max_tiles";
    return KS_ERR_POOL_FULL;
  }
  return KS_OK;
}

extern "C" {

int ks_default_config(ks_config* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  std::memset(c, 0, sizeof(*c));
  c->voxel_size = 0.05f;
  c->voxels_per_side = 16;
  c->truncation_distance = 4 * 0.05f;
  c->max_weight = 10000.0f;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->voxel_carving_enabled = 1;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->integration_order_mode = KS_ORDER_MIXED;
  c->integrator_threads = 1;
  c->method = KS_METHOD_FAST;
  c->bundle_order = KS_BUNDLE_ORDER_REFERENCE;
  c->semantic_measurement_probability = 0.9f;
  c->color_mode = KS_COLOR_MODE_SEMANTIC;
  c->n_dynamic_labels = 0;
  c->device_id = 0;
  c->max_tiles = 1u << 16;
  c->max_points = 1u << 20;
  return KS_OK;
}

int ks_create(const ks_config* cfg, ks_ctx** out) {
  if (!cfg || !out) {
    g_create_error = "null argument";
    return KS_ERR_INVALID_ARG;
  }
  const int vps = cfg->voxels_per_side;
  if (!(vps == 8 || vps == 16 || vps == 32 || vps == 64)) {
    g_create_error = "voxels_per_side must be 8, 16, 32 or 64";
    return KS_ERR_INVALID_ARG;
  }
  if (cfg->n_dynamic_labels < 0 || cfg->n_dynamic_labels > 32 || cfg->max_tiles == 0 || cfg->max_tiles >= (1u << 23)) {
    g_create_error = "bad n_dynamic_labels / max_tiles";
    return KS_ERR_INVALID_ARG;
  }
  // setSemanticProbabilities CHECKs [K:src/semantic_integrator_base.cpp:93-107]
  const float match = cfg->semantic_measurement_probability;
  const float non_match = 1.0f - cfg->semantic_measurement_probability;
  if (!(match > 0.0f) || !(non_match > 0.0f) || !(match < 1.0f) || !(non_match < 1.0f)) {
    g_create_error = "semantic_measurement_probability must be in (0,1)";
    return KS_ERR_PROBABILITY;
  }
  const float lm = std::log(match), lnm = std::log(non_match);  // host libm, as the reference
  if (!(lm > lnm)) {
    g_create_error = "log(p) must exceed log(1-p)";
    return KS_ERR_PROBABILITY;
  }
  if (cfg->early_out_phase_growth != 0 && cfg->early_out_phase_growth != KS_EARLY_OUT_EXACT &&
      (cfg->early_out_phase_growth < 16 || cfg->early_out_phase_growth > 4096)) {
    g_create_error = "early_out_phase_growth must be 0 / KS_EARLY_OUT_EXACT (the reference's serial result) or 16..4096 (ordered phases, growth in 1/16ths)";
    return KS_ERR_INVALID_ARG;
  }
  // the early-out can never fire if the threshold exceeds the longest possible ray
  const double max_steps = 3.0 * ((double)cfg->max_ray_length_m + 2.0 * cfg->truncation_distance) / (double)cfg->voxel_size + 8.0;
  const bool uses_early_out = (cfg->method == KS_METHOD_FAST) && ((double)cfg->max_consecutive_ray_collisions < max_steps);
  if (uses_early_out && (cfg->clear_checks_every_n_frames > 256)) {
    g_create_error = "fast integrator with the early-out enabled supports clear_checks_every_n_frames <= 256";
    return KS_ERR_UNSUPPORTED;
  }
  if (cfg->integration_order_mode != KS_ORDER_MIXED && cfg->integration_order_mode != KS_ORDER_SORTED &&
      cfg->integration_order_mode != KS_ORDER_MIXED_1024_GROUPS) {
    g_create_error = "integration_order_mode must be KS_ORDER_MIXED, KS_ORDER_SORTED or KS_ORDER_MIXED_1024_GROUPS";
    return KS_ERR_INVALID_ARG;
  }
  // (The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues, default 4, and kernels of streams that share a queue
  // run one after the other; a pipelined context keeps up to seven streams busy and runs best with 8.  That is the HOST
  // PROCESS's setting, read by the HIP runtime when it initialises: the library does not touch the environment —
  // INTEGRATION.md 4.2 says where the embedding process sets it; bench.py and the demos do so before their first HIP call.)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device_id >= ndev) {
    g_create_error = "no HIP device (the MI355X path has no CPU fallback)";
    return KS_ERR_NO_DEVICE;
  }
  ks_ctx* c = new ks_ctx();
  c->cfg = *cfg;
  const bool want_exact = cfg->early_out_phase_growth == 0 || cfg->early_out_phase_growth == KS_EARLY_OUT_EXACT;
  c->exact_early_out = uses_early_out && want_exact;
  if (want_exact) {
    // the seed's schedule.  Any seed gives the same map (the fixed point is unique); what a finer one buys is fewer marks to
    // emit, sort and evaluate, what it costs is k_test launches.  Measured at 640x480 / 5 cm on rings of 40 / 20 frames
    // (profiles/r05_c2_seed_growth_ab.txt), ms per frame: 32 (12 phases) 0.575 / 0.487, 28: 0.559, 26: 0.533 / 0.467,
    // 24: 0.541 / 0.461, 22 (23 phases): 0.523 / 0.465, 20: 0.525 / 0.484, 18: 0.558 / 0.544; coarser ones lose more (64: +12 %),
    // and so does a longest phase of 64 / 96 / 128 generations on top of any of them (equal steps at the end: +1 ... +8 %).
    c->cfg.early_out_phase_growth = 22;
    if (const char* sg = dbg_env("KS_EXACT_SEED_GROWTH")) c->cfg.early_out_phase_growth = std::min(4096, std::max(16, atoi(sg)));   // tuning runs
  }
  {
    const char* hl = dbg_env("KS_EXACT_HOST_LOOP");   // diagnostics / A-B: the host-driven fix-point loop of round 3 for every frame
    c->eo_device = c->exact_early_out && !(hl && hl[0] == '1');
    // The host-driven loop waits for the device per iteration: one frame at a time.  The event-driven fix point is
    // pipelined as long as a frame's marks are never seen by the next one (every frame bumps the set offset): what is left
    // of the dependence between frames — the zero-initialised slot — is carried by the commit events.
    if (c->exact_early_out && (!c->eo_device || c->cfg.clear_checks_every_n_frames > 1)) c->cfg.pipeline_frames = 0;
    const bool wide_rays = steps_max_of(c->cfg, (float)(1.0 / cfg->voxel_size)) > 400;
    // (long rays: a frame's marks cover whole rays — 2e8 marks of 33 bytes at 1280x720 / 2 cm — and a frame is tens of
    // milliseconds of GPU work: one frame at a time, one set of mark buffers)
    if (c->exact_early_out && wide_rays) c->cfg.pipeline_frames = 0;
    // (rounds the fix point needs from the doubling seed at 640x480 / 5 cm: 7 - 20 per frame over the bench trajectory, 9 - 12
    // on average.  A round that finds its list empty costs a launch of ~2 us; a round the ONE-workgroup finisher has to run in its
    // place costs ~0.1 ms: measured on a ring of 40 frames, 8 rounds as launches 0.681 ms/frame, 14 rounds 0.569
    // (profiles/r05_bulk_rounds_ab.txt).  A finisher that is handed too long a list asks for more: eo_want_bulk.)
    c->eo_bulk_rounds = wide_rays ? 32 : 20;
    if (const char* br = dbg_env("KS_EXACT_BULK_ROUNDS")) c->eo_bulk_rounds = std::min((int)kEoBulkMax, std::max(1, atoi(br)));
    if (const char* tr = dbg_env("KS_EXACT_TRACE")) c->eo_trace = tr[0] == '1';
    if (const char* sw = dbg_env("KS_EXACT_SWEEPS")) c->eo_sweeps = std::min(64, std::max(1, atoi(sw)));   // (tuning runs: any value gives the same map)
    if (const char* so = dbg_env("KS_EXACT_SWEEP_ORDER")) c->eo_sweep_order = so[0] == '0' ? 0 : 1;
  }
  c->uses_early_out = uses_early_out;
  c->use_bundle_rank = cfg->method == KS_METHOD_MERGED && cfg->bundle_order == KS_BUNDLE_ORDER_REFERENCE;
  if (const char* bh = dbg_env("KS_BO_HINT")) {   // tests / A-B: a fixed bundle-count hint (0: as many as the frame has points)
    c->bo_hint_fixed = true;
    c->bo_hint.store(atoi(bh) > 0 ? (uint32_t)atoi(bh) : ~0u, std::memory_order_relaxed);
  }
  if (cfg->method == KS_METHOD_MERGED && !cfg->enable_anti_grazing) {
    // stage A groups the points by end voxel: a 32-bit key (the voxel relative to a window around the sensor that holds every
    // point within max_ray; anything else through a small hash table) sorts in four passes instead of the 64-bit key's eight.
    // Not with anti-grazing (its binary search wants the 64-bit keys in order), not for windows wider than 10 bits per axis.
    const double extent = 2.0 * std::ceil(((double)cfg->max_ray_length_m + 2.0 * (double)cfg->voxel_size) / (double)cfg->voxel_size) + 8.0;
    unsigned w = 1;
    while (w < 32 && (double)(1u << w) < extent) ++w;
    if (const char* kb = dbg_env("KS_KEY_WINDOW_BITS")) w = (unsigned)std::max(0, atoi(kb));   // tests: 1..10 = a window that small (the overflow path), 0 = 64-bit keys
    c->key_bits = (w >= 1 && w <= 10) ? w : 0;
  }
  {
    const char* hpf = dbg_env("KS_HOST_PROF");
    c->host_prof = hpf && hpf[0] == '1';
    const char* ng = dbg_env("KS_NO_GRAPH");
    c->use_graphs = !(ng && ng[0] == '1');
    c->defer_join = true;
  }
  c->log_match = lm;
  c->log_non_match = lnm;
  c->voxel_size_inv = (float)(1.0 / cfg->voxel_size);  // TsdfIntegratorBase::setLayer
  c->vps_shift = vps == 8 ? 0 : vps == 16 ? 1 : vps == 32 ? 2 : 3;
#define CRCHK(expr)                                                        \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_);  \
      ks_destroy(c);                                                       \
      return KS_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)
  CRCHK(hipSetDevice(cfg->device_id));
  {
    // (experiment, KS_FRONT_PRIO = 1 / 2: the front stream at the runtime's high / low priority)
    const char* fp = dbg_env("KS_FRONT_PRIO");
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (fp && (fp[0] == '1' || fp[0] == '2')) CRCHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, fp[0] == '1' ? hi : lo));
    else CRCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  }
  // Frames in flight share nothing in stage B when a frame's early-out marks can never be seen by the next frame
  // (every frame bumps the set offset).  Then either (pipeline_frames < 8) every frame's stage B is its own launch
  // sequence and up to four of them run side by side on four streams, or (pipeline_frames = 8) stage B of four
  // consecutive frames is launched as ONE batch on one stream; each frame in flight has an early-out table of its
  // own.  Otherwise stage B stays strictly in frame order.
  const bool frames_independent = !uses_early_out || c->cfg.clear_checks_every_n_frames <= 1;
  c->batch = 1;
  if (const char* ov = dbg_env("KS_TEST_OVERLAP")) c->test_overlap = atoi(ov) != 0;   // diagnostics: the same schedule and result, rounds one after the other
  if (c->cfg.pipeline_frames >= 2 && frames_independent && (!c->exact_early_out || c->eo_device) && c->cfg.integration_order_mode != KS_ORDER_SORTED) {
    // (measured, 640x480: a batch of 4 behind 8 frames of lag ~ four single-frame sequences on four streams behind 4
    // frames of lag; batches of 2 or 3 lose to both: DESIGN.md)
    c->batch = c->cfg.pipeline_frames >= 16 ? 8 : c->cfg.pipeline_frames >= 8 ? 4 : 1;
    if (const char* bs = dbg_env("KS_BATCH")) c->batch = std::min(kBatchMax, std::max(1, atoi(bs)));  // diagnostics
  }
  // slots: the lag plus one batch being filled, a multiple of the batch (a batch then always starts on the same slots: its
  // captured launch sequence is found again)
  c->n_slots = !c->cfg.pipeline_frames ? 1 : (c->cfg.pipeline_frames > 8 || c->batch > 4) ? kSlots : 12;
  if (c->cfg.pipeline_frames) {
    // (shared early-out table: stage B of consecutive frames stays in order on one stream)
    c->n_march = (!frames_independent || c->batch > 1) ? 1 : std::min(kMarchStreams, std::max(4, c->cfg.pipeline_frames));
    // (exact early-out: a frame's stage B is a chain of ~75 small launches, ~1.5 ms long, and the hardware runs two or three
    // such chains side by side at best — measured: 8 streams lose to 4.  With pipeline_frames = 8 the chain is shared by
    // the four frames of a batch, and two batches alternate over two streams.)
    if (c->exact_early_out && c->n_march > 4) c->n_march = 4;
    if (c->exact_early_out && c->batch > 1) c->n_march = 2;
    if (const char* ms = dbg_env("KS_MARCH_STREAMS")) c->n_march = std::min(kMarchStreams, std::max(1, atoi(ms)));  // diagnostics
    {
      auto mk = [&](hipStream_t* st, char) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
      // without an early-out stage B is short (scan + emission): it follows stage A on the same stream, and the three
      // streams that remain (A+B, T, long runs) map onto hardware queues of their own
      if (!uses_early_out) {
        c->n_march = 1;
        c->stream_march_[0] = c->stream;
        const char* et = dbg_env("KS_EMIT_ON_TAIL");   // A/B: 0 = always after stage A on the front stream, 1 = always on the tail stream
        c->emit_on_tail = et && et[0] != '0';
        if (et && et[0] == '1') c->emit_on_tail_max_pairs = ~0ull;
      } else {
        for (int i = 0; i < c->n_march; ++i) CRCHK(mk(&c->stream_march_[i], 'm'));
      }
      {
        // (experiment, KS_TAIL_PRIO = 1 / 2: the tail stream at the runtime's high / low priority)
        const char* tp = dbg_env("KS_TAIL_PRIO");
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (tp && (tp[0] == '1' || tp[0] == '2')) CRCHK(hipStreamCreateWithPriority(&c->stream_tail, hipStreamNonBlocking, tp[0] == '1' ? hi : lo));
        else CRCHK(mk(&c->stream_tail, 't'));
      }
    }
  } else {
    c->stream_march_[0] = c->stream_tail = c->stream;
  }
  {
    // Created LAST.  The runtime spreads streams over its hardware queues in creation order, and kernels of streams that
    // share a hardware queue run one after the other: the heavy chains (stage B, stage T) must not share one.  With
    // the default of four hardware queues — one of which other streams of the process use — stage A and the long
    // runs (the two lightest: ~90 + ~65 us per 640x480 frame) are the pair that shares.
    const char* nl = dbg_env("KS_NO_LONG_STREAM");   // diagnostics: long runs on the tail stream, after k_apply
    if (nl && nl[0] == '1') c->stream_long = nullptr;
    else CRCHK(hipStreamCreateWithFlags(&c->stream_long, hipStreamNonBlocking));
    // the runs of more than kXLongRun updates (the voxels next to the sensor) on a stream of their own, four waves per run
    // (k_apply_xlong).  Same arithmetic, same order: the map does not change.  KS_XLONG=0 (diagnostics): one list, k_apply_long.
    if (const char* ar = dbg_env("KS_APPLY_RUNS")) c->apply_runs_min_pairs = atoi(ar) ? 0ull : ~0ull;   // tests / A-B: always / never
    const char* xp = dbg_env("KS_XLONG");
    c->xlong = xp ? atoi(xp) != 0 : true;
    if (c->xlong && c->stream_long) CRCHK(hipStreamCreateWithFlags(&c->stream_xlong, hipStreamNonBlocking));
    {
      // A/B only (KS_APPLY_STREAM=1): measured, it LOSES — C4-merged 6.12 vs 6.00 ms/frame, C3 0.75 vs 0.60 (one more stream for the
      // runtime's hardware queues to share) — so k_apply_runs stays on the tail stream
      const char* as = dbg_env("KS_APPLY_STREAM");
      if (c->stream_xlong && as && as[0] == '1') {
        // (experiment: which hardware queue / pipe the stream lands on — streams created and never used in front of it)
        if (const char* pad = dbg_env("KS_STREAM_PAD"))
          for (int i = 0; i < atoi(pad) && i < 8; ++i) {
            hipStream_t dummy = nullptr;
            CRCHK(hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking));
            c->stream_pad.push_back(dummy);
          }
        CRCHK(hipStreamCreateWithFlags(&c->stream_apply, hipStreamNonBlocking));
      }
    }
    if (const char* ll = dbg_env("KS_LONG_LANES")) {   // A/B: 0 = k_apply_long (two wavefronts per run) for all of them, 2 = lanes for frames of any size (tests)
      c->long_lanes = atoi(ll) != 0;
      if (atoi(ll) == 2) c->long_lanes_min_pairs = 0ull;
    }
    if (c->stream_long && c->long_lanes)
      for (int b = 0; b < 2; ++b) CRCHK(hipMalloc((void**)&c->d_long_hdr_[b], sizeof(LongHdr)));
    if (const char* lm = dbg_env("KS_LONG_MIN")) c->long_min_lanes = atoi(lm) == 16;
    {
      // merged, pipelined: the long bundles' merge beside the bundle order (integrate_device_impl) — on the stream of the long
      // voxel runs, which is idle four fifths of a 640x480 frame.  NOT on a stream of its own: a fifth active hardware queue
      // costs every other kernel of the front stream ~40 us (measured: C3 1.24 ms/frame instead of 0.62; the kernel trace shows
      // the bumps, with GPU_MAX_HW_QUEUES = 8 as with 4 — profiles/r06_bundle_stream_ab.txt).
      // KS_BUNDLE_STREAM = 0: in line, 1: its own stream (the A/B above)
      const char* bs = dbg_env("KS_BUNDLE_STREAM");
      const int mode = bs ? atoi(bs) : 0;
      if (c->cfg.method == KS_METHOD_MERGED && c->cfg.pipeline_frames && mode == 1) {
        // (experiment: which hardware queue / pipe the stream lands on — streams created and never used in front of it)
        if (const char* pad = dbg_env("KS_STREAM_PAD"))
          for (int i = 0; i < atoi(pad) && i < 8 && !c->stream_apply; ++i) {
            hipStream_t dummy = nullptr;
            CRCHK(hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking));
            c->stream_pad.push_back(dummy);
          }
        CRCHK(hipStreamCreateWithFlags(&c->stream_bundles_own, hipStreamNonBlocking));
      }
      if (c->cfg.method == KS_METHOD_MERGED && c->cfg.pipeline_frames && mode != 0)
        c->stream_bundles = c->stream_bundles_own ? c->stream_bundles_own : c->stream_long;
    }
    if (const char* sf = dbg_env("KS_SORT_FRONT")) c->sort_on_front = atoi(sf) != 0;
    if (const char* ld = dbg_env("KS_LANES_DEPTH")) c->lanes_depth = atoi(ld) == 4 ? 4u : 6u;
    if (const char* rt = dbg_env("KS_RUN_THREADS")) c->run_threads = atoi(rt) == 512 ? 512u : 256u;
    if (const char* xl = dbg_env("KS_XL_PARALLEL")) {   // A/B: 0 = every such run through k_apply_xlong, 2 = the integer-sum path for frames of any size (tests)
      c->xl_parallel = atoi(xl) != 0;
      if (atoi(xl) == 2) c->xl_min_pairs = 0ull;
    }   // A/B: 0 = every such run through k_apply_xlong
    if (c->stream_xlong && c->xl_parallel) {
      CRCHK(hipMalloc((void**)&c->d_xl_runs, kXlMaxRuns * sizeof(XlRun)));
      CRCHK(hipMalloc((void**)&c->d_xl_idx, kXlMaxRuns * sizeof(uint32_t)));
      CRCHK(hipMalloc((void**)&c->d_xl_hdr, sizeof(XlHeader)));
      CRCHK(hipMemset(c->d_xl_hdr, 0, sizeof(XlHeader)));
      CRCHK(hipMalloc((void**)&c->d_xl_chunks, (size_t)c->cap_xl_chunks * sizeof(XlChunk)));
    }
  }
  for (auto& P : c->pset) {
    for (auto& e : P.ev) CRCHK(hipEventCreate(&e));
    CRCHK(hipEventCreate(&P.k0));
    CRCHK(hipEventCreate(&P.k1));
  }
  uint32_t cap = 1024;
  while (cap < 2u * cfg->max_tiles) cap <<= 1;
  c->table.mask = cap - 1;
  c->table.max_tiles = cfg->max_tiles;
  const size_t mt = cfg->max_tiles;
  CRCHK(hipMalloc((void**)&c->table.ent, cap * sizeof(TileEntry)));
  CRCHK(hipMalloc((void**)&c->table.slot_keys, mt * sizeof(uint64_t)));
  CRCHK(hipMemset(c->table.ent, 0xff, cap * sizeof(TileEntry)));  // key = empty, val = kSlotPending
  CRCHK(hipMalloc((void**)&c->pool.vox, mt * kTileVoxels * 8 * sizeof(uint4)));
  CRCHK(hipMalloc((void**)&c->pool.updated, mt));
  CRCHK(hipMemset(c->pool.updated, 0, mt));
  CRCHK(hipMalloc((void**)&c->pool.dirty, mt));
  CRCHK(hipMemset(c->pool.dirty, 0, mt));
  CRCHK(hipMalloc((void**)&c->d_start_set, sizeof(uint64_t) << kSetBits));
  c->n_obs = (uses_early_out && frames_independent) ? std::min(kObsTables, std::max(c->n_march, c->batch * c->n_march)) : 1;
  for (int t = 0; t < c->n_obs; ++t) {
    CRCHK(hipMalloc((void**)&c->d_observed_[t], 2 * (sizeof(uint64_t) << kSetBits)));   // {newest, older} per slot
    CRCHK(hipMemset(c->d_observed_[t], 0, 2 * (sizeof(uint64_t) << kSetBits)));
    CRCHK(hipMemcpy(c->d_observed_[t], &kObsPoison, 8, hipMemcpyHostToDevice));
  }
  CRCHK(hipMemset(c->d_start_set, 0, sizeof(uint64_t) << kSetBits));
  const uint64_t poison = ~0ull;  // ApproxHashSet ctor: slot[offset_=0] = SIZE_MAX
  CRCHK(hipMemcpy(c->d_start_set, &poison, 8, hipMemcpyHostToDevice));
  CRCHK(hipMalloc((void**)&c->d_retry_counters, sizeof(Counters)));
  if (c->exact_early_out) {
    CRCHK(hipMalloc((void**)&c->d_eo_committed, 64));
    CRCHK(hipMemset(c->d_eo_committed, 0, 64));
    CRCHK(hipMalloc((void**)&c->d_eo_range, sizeof(uint2) << kSetBits));
    CRCHK(hipMalloc((void**)&c->d_eo_plain, sizeof(uint64_t) << kSetBits));
    CRCHK(hipMemset(c->d_eo_plain, 0, sizeof(uint64_t) << kSetBits));
    CRCHK(hipMemcpy(c->d_eo_plain, &poison, 8, hipMemcpyHostToDevice));  // ApproxHashSet ctor: slot[0] = SIZE_MAX
    CRCHK(hipMalloc((void**)&c->d_eo_state, sizeof(EoState)));
    CRCHK(hipHostMalloc((void**)&c->h_eo_state, sizeof(EoState)));
  }
  CRCHK(hipMalloc((void**)&c->d_label_lut, 256 * sizeof(uint32_t)));
  CRCHK(hipMemcpy(c->d_label_lut, cfg->label_rgba, 1024, hipMemcpyHostToDevice));
  static_assert(sizeof(Counters) == 32, "snapshot layout");
  CRCHK(hipMalloc((void**)&c->d_state, 64 * (kSlots + 1)));
  CRCHK(hipMemset(c->d_state, 0, 64 * (kSlots + 1)));
  c->table.n_tiles = (uint32_t*)(c->d_state + 64 * kSlots);
  for (int i = 0; i < kSlots; ++i) {
    FrameSlot& S = c->slot[i];
    S.index = i;
    S.d_counters = (Counters*)(c->d_state + 64 * i);
    CRCHK(hipHostMalloc((void**)&S.h_snap, sizeof(HostSnap)));
    CRCHK(hipMalloc((void**)&S.d_F, sizeof(FrameParams)));
    std::memset(S.h_snap, 0, sizeof(HostSnap));
    CRCHK(hipEventCreateWithFlags(&S.a_done, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.bl_fork, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.bl_join, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.tail_done, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.join, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.join_x, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.found, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.applied, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&S.sorted, hipEventDisableTiming));
  }
#undef CRCHK
  // pair buffers start at 4 updates per point of the largest cloud (a frame that needs more grows its buffer and
  // repeats the emission once)
  c->pairs_hint.store((size_t)cfg->max_points * 4, std::memory_order_relaxed);
  size_t eo_marks0 = std::max<size_t>((size_t)1 << 20, 4 * (size_t)cfg->max_points), eo_x0 = std::max<size_t>((size_t)1 << 17, (size_t)cfg->max_points);
  // long rays: the marks cover whole rays (measured at 1280x720 / 2 cm / 10 m: 250 marks per point of the cloud; room for a third of
  // the longest possible ray = ~10 GB there).  Asked for here, allocated by the first integrate call (the growth path at the top of
  // integrate_device_impl, before the frame is enqueued): a context that is created and never integrates (a map that is only loaded, merged into
  // or read back) does not hold it, and a device without the room falls back to the host-driven loop instead of failing
  // ks_create.
  size_t eo_marks_first = 0;
  if (c->exact_early_out && c->eo_device && steps_max_of(c->cfg, c->voxel_size_inv) > 400)
    eo_marks_first = (size_t)cfg->max_points * (steps_max_of(c->cfg, c->voxel_size_inv) / 3 + 32);
  // (tests: start small, so that the overflow -> host-driven loop -> grow path is exercised)
  if (const char* e = dbg_env("KS_EXACT_CAP_MARKS")) eo_marks0 = std::max<size_t>(64, (size_t)atoll(e)), eo_marks_first = 0;
  if (const char* e = dbg_env("KS_EXACT_CAP_X")) eo_x0 = std::max<size_t>(8, (size_t)atoll(e));
  if (eo_marks_first > eo_marks0) c->eo_want_marks.store(eo_marks_first, std::memory_order_relaxed);
  if (ensure_points(c, cfg->max_points) != KS_OK || ensure_exact_slots(c, eo_marks0, eo_x0) != KS_OK) {
    g_create_error = c->err;
    ks_destroy(c);
    return KS_ERR_HIP;
  }
  {
    const char* nt = dbg_env("KS_NO_TAIL_THREAD");
    c->use_tail_thread = c->cfg.pipeline_frames > 0 && !(nt && nt[0] == '1');
    if (c->use_tail_thread) c->tail_thread = std::thread(tail_worker, c);
  }
  *out = c;
  return KS_OK;
}

void ks_destroy(ks_ctx* c) {
  if (!c) return;
  if (c->tail_thread.joinable()) {
    {
      std::lock_guard<std::mutex> lk(c->tail_mu);
      c->tail_quit = true;
    }
    c->tail_cv.notify_all();
    c->tail_thread.join();
  }
  if (c->host_prof && c->frame_no)
    fprintf(stderr, "[ks host prof] frames %llu: per frame us  A %.1f  B %.1f  T %.1f  (radix sort launches inside A+T: %.1f)\n",
            (unsigned long long)c->frame_no, 1e6 * (c->hp_a - c->hp_b) / c->frame_no, 1e6 * c->hp_b / c->frame_no,
            1e6 * c->hp_t / c->frame_no, 1e6 * c->hp_sort / c->frame_no);
  if (c->stream_tail && c->stream_tail != c->stream) (void)hipStreamSynchronize(c->stream_tail);
  for (auto sm : c->stream_march_)
    if (sm && sm != c->stream) (void)hipStreamSynchronize(sm);
  if (c->stream_long) (void)hipStreamSynchronize(c->stream_long);
  if (c->stream_bundles_own) (void)hipStreamSynchronize(c->stream_bundles_own);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->table.ent, c->table.slot_keys, c->pool.vox, c->pool.updated, c->pool.dirty, c->d_start_set, c->d_observed_[0], c->d_observed_[1], c->d_observed_[2], c->d_observed_[3], c->d_observed_[4], c->d_observed_[5], c->d_observed_[6], c->d_observed_[7], c->d_observed_[8], c->d_observed_[9], c->d_observed_[10], c->d_observed_[11], c->d_observed_[12], c->d_observed_[13], c->d_observed_[14], c->d_observed_[15], c->d_color_lut,
                  c->d_label_lut, c->d_xyz, c->d_rgba, c->d_labels, c->d_hash, c->d_skeys32, c->d_skeys32b, c->d_gpw, c->d_glc, c->d_ray_keys, c->d_long_list_[0], c->d_long_list_[1], c->d_blong, c->d_blong_merged, c->d_key_overflow, c->d_pkeys,
                  c->d_pkeys2, c->d_pvals, c->d_pvals2, c->d_order, c->d_inv_order, c->d_okeys, c->d_okeys2, c->d_ovals,
                  c->d_pairs2_[0], c->d_pairs2_[1], c->d_state, c->d_xchg_u32, c->d_xchg_u64, c->d_retry_counters,
                  c->d_block_idx, c->d_tsdf_out, c->d_sem_out, c->d_vox_out, c->d_depth_blocks, c->d_img_depth, c->d_img_aux, c->d_bo_slab,
                  c->d_eo_keys[0], c->d_eo_keys[1], c->d_eo_vals[0], c->d_eo_vals[1], c->d_eo_range, c->d_eo_plain, c->d_eo_lp, c->d_eo_bt,
                  c->d_eo_state, c->d_xl_runs, c->d_xl_hdr, c->d_xl_chunks, c->d_xl_idx, c->d_xl_fb, c->d_long_sorted_[0], c->d_long_sorted_[1], c->d_long_hdr_[0], c->d_long_hdr_[1], c->d_sh_okey[0], c->d_sh_okey[1], c->d_sh_gkey[0], c->d_sh_gkey[1], c->d_sh_seq[0], c->d_sh_seq[1], c->d_sh_sdf[0], c->d_sh_sdf[1], c->d_sh_uw[0], c->d_sh_uw[1], c->d_sh_counts, c->d_sh_tk, c->d_sh_pairs[0], c->d_sh_pairs[1], c->d_sh_vals[0], c->d_sh_vals[1], c->d_rx_counts, c->d_tx_keys, c->d_rx_keys, c->d_tx_slots, c->d_tx_payload, c->d_rx_payload};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& S : c->slot) {
    for (void* p : {(void*)S.d_rays, (void*)S.d_deltas, (void*)S.d_ray_list, (void*)S.d_pairs, (void*)S.d_cnt, (void*)S.d_lp,
                    (void*)S.d_bt, (void*)S.d_live, (void*)S.d_F, (void*)S.d_gkeys, (void*)S.d_rkeys,
                    (void*)S.d_eo_keys[0], (void*)S.d_eo_keys[1], (void*)S.d_eo_vals[0], (void*)S.d_eo_vals[1], (void*)S.d_eo_tab, (void*)S.d_eo_xnode,
                    (void*)S.d_eo_cnt_b, (void*)S.d_eo_ux, (void*)S.d_eo_dirty, (void*)S.d_eo_list[0], (void*)S.d_eo_list[1], (void*)S.d_eo_chg,
                    (void*)S.d_eo_consulted, (void*)S.d_eo_lp, (void*)S.d_eo_bt, (void*)S.d_eo_ctl, (void*)S.d_eo_sort_ws, (void*)S.d_eo_hitb, (void*)S.d_eo_bits_a, (void*)S.d_eo_bits_b, (void*)S.d_eo_btp, (void*)S.d_eo_hseq, (void*)S.d_eo_where, (void*)S.d_eo_rinfo, (void*)S.d_eo_ckpt})
      if (p) (void)hipFree(p);
    if (S.eo_committed) (void)hipEventDestroy(S.eo_committed);
  }
  if (c->d_eo_committed) (void)hipFree(c->d_eo_committed);
  if (c->h_eo_state) (void)hipHostFree(c->h_eo_state);
  ksrs::release(c->sort_ws);
  ksrs::release(c->sort_ws_tail);
  for (auto& S : c->slot) {
    for (auto& G : S.b_graphs)
      for (hipGraphExec_t g : {G.g1, G.g2, G.g3})
        if (g) (void)hipGraphExecDestroy(g);
    if (S.h_snap) (void)hipHostFree(S.h_snap);
    if (S.ready) (void)hipEventDestroy(S.ready);
    if (S.tail_done) (void)hipEventDestroy(S.tail_done);
    if (S.fork) (void)hipEventDestroy(S.fork);
    if (S.join) (void)hipEventDestroy(S.join);
    if (S.join_x) (void)hipEventDestroy(S.join_x);
    if (S.found) (void)hipEventDestroy(S.found);
    if (S.applied) (void)hipEventDestroy(S.applied);
    if (S.sorted) (void)hipEventDestroy(S.sorted);
    if (S.a_done) (void)hipEventDestroy(S.a_done);
    if (S.bl_fork) (void)hipEventDestroy(S.bl_fork);
    if (S.bl_join) (void)hipEventDestroy(S.bl_join);
  }
  for (auto& P : c->pset) {
    for (auto& e : P.ev)
      if (e) (void)hipEventDestroy(e);
    if (P.k0) (void)hipEventDestroy(P.k0);
    if (P.k1) (void)hipEventDestroy(P.k1);
  }
  if (c->stream_tail && c->stream_tail != c->stream) (void)hipStreamDestroy(c->stream_tail);
  for (auto sm : c->stream_march_)
    if (sm && sm != c->stream) (void)hipStreamDestroy(sm);
  if (c->stream_long) (void)hipStreamDestroy(c->stream_long);
  if (c->stream_xlong) (void)hipStreamDestroy(c->stream_xlong);
  if (c->stream_apply) (void)hipStreamDestroy(c->stream_apply);
  if (c->stream_bundles_own) (void)hipStreamDestroy(c->stream_bundles_own);
  for (hipStream_t d : c->stream_pad) (void)hipStreamDestroy(d);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* ks_last_error(ks_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ks_set_color_to_label(ks_ctx* c, const uint8_t* rgba_keys, const uint8_t* labels, size_t n) {
  if (!c || (n && (!rgba_keys || !labels))) return KS_ERR_INVALID_ARG;
  // 16 MiB direct-mapped rgb -> label table; lookups force alpha = 255
  // ([K:src/semantic_tsdf_integrator_fast.cpp:157]), so only keys with alpha 255 can match
  // (HashableColor::operator== compares alpha, [K:src/color.cpp:25-27]); unknown -> 0.
  std::vector<uint8_t> lut(1u << 24, 0);
  for (size_t i = 0; i < n; ++i) {
    if (rgba_keys[4 * i + 3] != 255) continue;
    const uint32_t rgb = rgba_keys[4 * i] | (rgba_keys[4 * i + 1] << 8) | (rgba_keys[4 * i + 2] << 16);
    lut[rgb] = labels[i];
  }
  if (int rc = quiesce(c)) return rc;  // frames in flight still read the old table
  if (!c->d_color_lut) HIPCHK(c, hipMalloc((void**)&c->d_color_lut, 1u << 24));
  HIPCHK(c, hipMemcpy(c->d_color_lut, lut.data(), 1u << 24, hipMemcpyHostToDevice));
  return KS_OK;
}

int ks_integrate_points_device(ks_ctx* c, const float T[7], const float* d_xyz, const uint8_t* d_rgba,
                               const uint8_t* d_labels, size_t n, int freespace, ks_frame_stats* stats) {
  if (!c || !T || (n && !d_xyz)) return KS_ERR_INVALID_ARG;
  if (!d_labels && !(d_rgba && c->d_color_lut)) {
    c->err = "labels == NULL requires rgba and a colour map (ks_set_color_to_label)";
    return KS_ERR_INVALID_ARG;
  }
  return integrate_device(c, T, d_xyz, d_rgba, d_labels, n, freespace, stats);
}

int ks_integrate_points(ks_ctx* c, const float T[7], const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                        size_t n, int freespace, ks_frame_stats* stats) {
  if (!c || !T || (n && !xyz)) return KS_ERR_INVALID_ARG;
  if (!labels && !(rgba && c->d_color_lut)) {
    c->err = "labels == NULL requires rgba and a colour map (ks_set_color_to_label)";
    return KS_ERR_INVALID_ARG;
  }
  int rc;
  if (n > c->cap_points && (rc = quiesce(c))) return rc;  // growing frees buffers a pending tail still needs
  if ((rc = ensure_points(c, n))) return rc;
  if (n) {
    HIPCHK(c, hipMemcpyAsync(c->d_xyz, xyz, n * 12, hipMemcpyHostToDevice, c->stream));
    if (rgba) HIPCHK(c, hipMemcpyAsync(c->d_rgba, rgba, n * 4, hipMemcpyHostToDevice, c->stream));
    if (labels) HIPCHK(c, hipMemcpyAsync(c->d_labels, labels, n, hipMemcpyHostToDevice, c->stream));
  }
  return integrate_device(c, T, c->d_xyz, rgba ? c->d_rgba : nullptr, labels ? c->d_labels : nullptr, n, freespace, stats);
}

// n_known: the number of valid pixels if the caller has counted them (the host-pointer entry: the image is in host memory
// anyway), else -1: the compacted count is read back from the device before the frame is enqueued.
static int integrate_depth_impl(ks_ctx* c, const float T[7], DepthParams D, int freespace, ks_frame_stats* stats, long long n_known = -1) {
  const size_t n_px = (size_t)D.width * D.height;
  int rc;
  if (n_px > c->cap_points && (rc = quiesce(c))) return rc;  // growing frees buffers a pending tail still needs
  if ((rc = ensure_points(c, n_px))) return rc;
  const uint32_t nb = (uint32_t)((n_px + 1023) / 1024);
  if (nb + 1 > c->cap_depth_blocks) {
    if ((rc = dev_alloc(c, &c->d_depth_blocks, (size_t)nb + 1))) return rc;
    c->cap_depth_blocks = nb + 1;
  }
  hipStream_t st = c->stream;
  hipLaunchKernelGGL(k_depth_count, dim3(nb), dim3(1024), 0, st, D, (uint32_t)n_px, c->d_depth_blocks);
  hipLaunchKernelGGL(k_depth_scan, dim3(1), dim3(1024), 0, st, c->d_depth_blocks, nb);
  const bool have_labels = D.label_img != nullptr;
  hipLaunchKernelGGL(k_depth_compact, dim3(nb), dim3(1024), 0, st, D, (uint32_t)n_px, c->d_depth_blocks, c->d_label_lut,
                     c->d_xyz, c->d_rgba, c->d_labels);
  uint32_t n = 0;
  if (n_known >= 0) {
    n = (uint32_t)n_known;   // (the integration order and the sort-key layout depend on n: the host needs it to enqueue the frame)
  } else {
    HIPCHK(c, hipMemcpyAsync(&n, c->d_depth_blocks + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  if (!have_labels && !(D.rgba_img && c->d_color_lut)) {
    c->err = "ks_integrate_depth: need a label image, or a colour image plus ks_set_color_to_label";
    return KS_ERR_INVALID_ARG;
  }
  return integrate_device(c, T, c->d_xyz, c->d_rgba, have_labels ? c->d_labels : nullptr, n, freespace, stats);
}

int ks_integrate_depth(ks_ctx* c, const float T[7], const void* depth, int depth_fmt, const uint8_t* label_img,
                       const uint8_t* rgba_img, int width, int height, const float K[4], int freespace,
                       ks_frame_stats* stats) {
  if (!c || !T || !depth || !K || width <= 0 || height <= 0 || (depth_fmt != 0 && depth_fmt != 1)) return KS_ERR_INVALID_ARG;
  const size_t n_px = (size_t)width * height;
  const size_t dbytes = n_px * (depth_fmt == 0 ? 4 : 2);
  if (dbytes > c->cap_img_depth) {
    int rc = dev_alloc(c, &c->d_img_depth, dbytes);
    if (rc) return rc;
    c->cap_img_depth = dbytes;
  }
  if (n_px * 4 > c->cap_img_aux) {
    int rc = dev_alloc(c, &c->d_img_aux, n_px * 4);
    if (rc) return rc;
    c->cap_img_aux = n_px * 4;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_img_depth, depth, dbytes, hipMemcpyHostToDevice, c->stream));
  DepthParams D{};
  D.depth = c->d_img_depth;
  if (label_img) {
    HIPCHK(c, hipMemcpyAsync(c->d_img_aux, label_img, n_px, hipMemcpyHostToDevice, c->stream));
    D.label_img = c->d_img_aux;
  } else if (rgba_img) {
    HIPCHK(c, hipMemcpyAsync(c->d_img_aux, rgba_img, n_px * 4, hipMemcpyHostToDevice, c->stream));
    D.rgba_img = c->d_img_aux;
  }
  D.fmt = depth_fmt;
  D.width = width;
  D.height = height;
  D.cx = K[2];
  D.cy = K[3];
  // unit_scaling / fx evaluated in double then narrowed, as depth_map_to_pointcloud.h:227-229
  const double unit = depth_fmt == 0 ? 1.0 : 0.001;
  D.constant_x = (float)(unit / (double)K[0]);
  D.constant_y = (float)(unit / (double)K[1]);
  // the valid pixels (ks_k_io.h: depth_pixel — finite f32 / non-zero u16), counted here while the copies are in flight: no
  // read-back of the compacted count, no host wait before the frame is enqueued
  long long n_valid = 0;
  if (depth_fmt == 0) {
    const float* d = (const float*)depth;
    for (size_t i = 0; i < n_px; ++i) n_valid += std::isfinite(d[i]) ? 1 : 0;
  } else {
    const uint16_t* d = (const uint16_t*)depth;
    for (size_t i = 0; i < n_px; ++i) n_valid += d[i] != 0 ? 1 : 0;
  }
  return integrate_depth_impl(c, T, D, freespace, stats, n_valid);
}

int ks_integrate_depth_device(ks_ctx* c, const float T[7], const void* d_depth, int depth_fmt, const uint8_t* d_label_img,
                              const uint8_t* d_rgba_img, int width, int height, const float K[4], int freespace,
                              ks_frame_stats* stats) {
  if (!c || !T || !d_depth || !K || width <= 0 || height <= 0 || (depth_fmt != 0 && depth_fmt != 1)) return KS_ERR_INVALID_ARG;
  DepthParams D{};
  D.depth = d_depth;
  D.label_img = d_label_img;
  D.rgba_img = d_label_img ? nullptr : d_rgba_img;
  D.fmt = depth_fmt;
  D.width = width;
  D.height = height;
  D.cx = K[2];
  D.cy = K[3];
  const double unit = depth_fmt == 0 ? 1.0 : 0.001;
  D.constant_x = (float)(unit / (double)K[0]);
  D.constant_y = (float)(unit / (double)K[1]);
  return integrate_depth_impl(c, T, D, freespace, stats);
}

int ks_num_blocks(ks_ctx* c, size_t* n) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  std::vector<int32_t> v;
  int rc = collect_block_indices(c, false, false, &v);
  if (rc) return rc;
  *n = v.size() / 3;
  return KS_OK;
}

int ks_get_block_indices(ks_ctx* c, int32_t* out, size_t cap, size_t* n) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  std::vector<int32_t> v;
  int rc = collect_block_indices(c, false, false, &v);
  if (rc) return rc;
  *n = v.size() / 3;
  if (out) std::memcpy(out, v.data(), std::min(cap, *n) * 3 * sizeof(int32_t));
  return KS_OK;
}

int ks_get_updated_block_indices(ks_ctx* c, int32_t* out, size_t cap, size_t* n, int reset) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  std::vector<int32_t> v;
  int rc = collect_block_indices(c, true, reset != 0, &v);
  if (rc) return rc;
  *n = v.size() / 3;
  if (out) std::memcpy(out, v.data(), std::min(cap, *n) * 3 * sizeof(int32_t));
  return KS_OK;
}

int ks_download_blocks(ks_ctx* c, const int32_t* idx, size_t n, void* tsdf_out, void* sem_out) {
  if (!c || (n && !idx)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (int rc = quiesce(c)) return rc;
  const int vps = c->cfg.voxels_per_side;
  const size_t nv = (size_t)vps * vps * vps;
  // chunk so staging buffers stay bounded (<= ~256 MiB of semantic voxels)
  const size_t chunk = std::max<size_t>(1, (size_t(256) << 20) / (nv * 92));
  if (c->cap_out_blocks < std::min(chunk, n)) {
    const size_t cb = std::min(chunk, std::max<size_t>(n, 16));
    int rc;
    if ((rc = dev_alloc(c, &c->d_tsdf_out, cb * nv * 12))) return rc;
    if ((rc = dev_alloc(c, &c->d_sem_out, cb * nv * 92))) return rc;
    if ((rc = dev_alloc(c, &c->d_block_idx, cb * 3))) return rc;
    c->cap_out_blocks = cb;
  }
  for (size_t off = 0; off < n; off += c->cap_out_blocks) {
    const size_t m = std::min(c->cap_out_blocks, n - off);
    HIPCHK(c, hipMemcpyAsync(c->d_block_idx, idx + 3 * off, m * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_download, dim3((uint32_t)((nv + 255) / 256), (uint32_t)m), dim3(256), 0, c->stream, c->table,
                       c->pool, c->d_block_idx, vps, c->d_label_lut, tsdf_out ? c->d_tsdf_out : nullptr,
                       sem_out ? c->d_sem_out : nullptr);
    if (tsdf_out)
      HIPCHK(c, hipMemcpyAsync((uint8_t*)tsdf_out + off * nv * 12, c->d_tsdf_out, m * nv * 12, hipMemcpyDeviceToHost, c->stream));
    if (sem_out)
      HIPCHK(c, hipMemcpyAsync((uint8_t*)sem_out + off * nv * 92, c->d_sem_out, m * nv * 92, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return KS_OK;
}

int ks_upload_blocks(ks_ctx* c, const int32_t* idx, size_t n, const void* tsdf_in, const void* sem_in) {
  if (!c || (n && !idx) || (!tsdf_in && !sem_in)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (c->fatal) return KS_ERR_INVALID_ARG;
  const int vps = c->cfg.voxels_per_side;
  const int tpb = vps / 8;  // tiles per block edge
  const size_t nv = (size_t)vps * vps * vps;
  std::vector<uint64_t> keys;
  keys.reserve(n * tpb * tpb * tpb);
  const int lim = kTileBias / tpb;
  for (size_t b = 0; b < n; ++b) {
    const int bx = idx[3 * b], by = idx[3 * b + 1], bz = idx[3 * b + 2];
    if (bx < -lim || bx >= lim || by < -lim || by >= lim || bz < -lim || bz >= lim) {
      c->err = "block index outside the packed tile-key range";
      return KS_ERR_INDEX_RANGE;
    }
    for (int z = 0; z < tpb; ++z)
      for (int y = 0; y < tpb; ++y)
        for (int x = 0; x < tpb; ++x) keys.push_back(pack_tile(bx * tpb + x, by * tpb + y, bz * tpb + z));
  }
  uint64_t* d_keys = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_keys, keys.size() * sizeof(uint64_t)));
  HIPCHK(c, hipMemcpyAsync(d_keys, keys.data(), keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  int rc = insert_tiles(c, d_keys, keys.size());
  (void)hipFree(d_keys);
  if (rc) return rc;
  // staging buffers shared with ks_download_blocks
  const size_t chunk = std::max<size_t>(1, (size_t(256) << 20) / (nv * 92));
  if (c->cap_out_blocks < std::min(chunk, n)) {
    const size_t cb = std::min(chunk, std::max<size_t>(n, 16));
    if ((rc = dev_alloc(c, &c->d_tsdf_out, cb * nv * 12))) return rc;
    if ((rc = dev_alloc(c, &c->d_sem_out, cb * nv * 92))) return rc;
    if ((rc = dev_alloc(c, &c->d_block_idx, cb * 3))) return rc;
    c->cap_out_blocks = cb;
  }
  for (size_t off = 0; off < n; off += c->cap_out_blocks) {
    const size_t m = std::min(c->cap_out_blocks, n - off);
    HIPCHK(c, hipMemcpyAsync(c->d_block_idx, idx + 3 * off, m * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (tsdf_in)
      HIPCHK(c, hipMemcpyAsync(c->d_tsdf_out, (const uint8_t*)tsdf_in + off * nv * 12, m * nv * 12, hipMemcpyHostToDevice, c->stream));
    if (sem_in)
      HIPCHK(c, hipMemcpyAsync(c->d_sem_out, (const uint8_t*)sem_in + off * nv * 92, m * nv * 92, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_upload, dim3((uint32_t)((nv + 255) / 256), (uint32_t)m), dim3(256), 0, c->stream, c->table, c->pool,
                       c->d_block_idx, vps, tsdf_in ? (const uint8_t*)c->d_tsdf_out : nullptr,
                       sem_in ? (const uint8_t*)c->d_sem_out : nullptr);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, hipGetLastError());
  return KS_OK;
}

// ---- voxel-level host sync -------------------------------------------------------------------------------
// Device-side address of a host allocation the GPU can write (hipHostMalloc'ed: ks_host_alloc), else nullptr.
static void* device_view_of_pinned(void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return a.type == hipMemoryTypeHost ? a.devicePointer : nullptr;
}

static int updated_voxels_impl(ks_ctx* c, void* out, size_t cap, size_t* n, bool count_only, ks_voxel_run* runs, size_t cap_runs,
                               size_t* n_runs) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  *n = 0;
  if (n_runs) *n_runs = 0;
  if (int rc = quiesce(c)) return rc;
  const uint32_t nt = c->tiles_initialised;
  if (nt == 0) return KS_OK;
  int rc;
  if ((rc = ensure_exchange(c, (size_t)nt + 8))) return rc;
  uint32_t* d_cnt = c->d_xchg_u32;          // [0] listed tiles, [1] dirty voxels
  uint32_t* d_list = c->d_xchg_u32 + 8;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemsetAsync(d_cnt, 0, 8 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_list_updated_tiles, dim3((nt + 255) / 256), dim3(256), 0, st, c->pool, nt, d_list, d_cnt);
  uint32_t h_cnt[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(h_cnt, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const uint32_t n_list = h_cnt[0];
  if (n_list == 0) return KS_OK;
  if (n_runs) *n_runs = n_list;
  hipLaunchKernelGGL(k_export_dirty, dim3(n_list), dim3(512), 0, st, c->table, c->pool, (const uint32_t*)d_list,
                     (const uint32_t*)c->d_label_lut, c->vps_shift, 1, d_cnt, (uint8_t*)nullptr, (uint32_t*)nullptr);
  HIPCHK(c, hipMemcpyAsync(h_cnt, d_cnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *n = h_cnt[1];
  if (count_only || h_cnt[1] == 0) return KS_OK;
  if ((size_t)h_cnt[1] > cap) {
    c->err = "ks_download_updated_voxels: output buffer too small (call ks_count_updated_voxels first)";
    return KS_ERR_INVALID_ARG;
  }
  if (runs && (size_t)n_list > cap_runs) {
    c->err = "ks_download_updated_voxels: run buffer too small";
    return KS_ERR_INVALID_ARG;
  }
  // pinned host targets (ks_host_alloc) are written by the kernel itself: no staging copy
  uint8_t* rec_direct = (uint8_t*)device_view_of_pinned(out);
  uint32_t* run_direct = runs ? (uint32_t*)device_view_of_pinned(runs) : nullptr;
  const size_t bytes = (size_t)h_cnt[1] * kVoxRecBytes;
  const size_t run_bytes = (size_t)n_list * sizeof(ks_voxel_run);
  const size_t need = (rec_direct ? 0 : bytes + bytes / 4) + (run_direct ? 0 : 2 * run_bytes) + 64;
  if ((!rec_direct || (runs && !run_direct)) && need > c->cap_vox_out) {
    if ((rc = dev_alloc(c, &c->d_vox_out, need))) return rc;
    c->cap_vox_out = need;
  }
  uint8_t* d_rec = rec_direct ? rec_direct : c->d_vox_out;
  uint32_t* d_runs = run_direct ? run_direct : (uint32_t*)(c->d_vox_out + (rec_direct ? 0 : ((bytes + 15) & ~(size_t)15)));
  HIPCHK(c, hipMemsetAsync(d_cnt + 1, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_export_dirty, dim3(n_list), dim3(512), 0, st, c->table, c->pool, (const uint32_t*)d_list,
                     (const uint32_t*)c->d_label_lut, c->vps_shift, 0, d_cnt, d_rec, runs ? d_runs : (uint32_t*)nullptr);
  if (!rec_direct) HIPCHK(c, hipMemcpyAsync(out, c->d_vox_out, bytes, hipMemcpyDeviceToHost, st));
  if (runs && !run_direct) HIPCHK(c, hipMemcpyAsync(runs, d_runs, run_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipGetLastError());
  return KS_OK;
}

int ks_count_updated_voxels(ks_ctx* c, size_t* n, size_t* n_runs) {
  return updated_voxels_impl(c, nullptr, 0, n, true, nullptr, 0, n_runs);
}
int ks_download_updated_voxels(ks_ctx* c, void* out, size_t cap, size_t* n, ks_voxel_run* runs, size_t cap_runs, size_t* n_runs) {
  if (!out && cap) return KS_ERR_INVALID_ARG;
  static_assert(sizeof(ks_voxel_run) == 20, "run record layout");
  return updated_voxels_impl(c, out, cap, n, false, runs, cap_runs, n_runs);
}

void* ks_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void ks_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int ks_debug_radix_sort(ks_ctx* c, void* keys, uint32_t* vals, size_t n, int key_bits, unsigned end_bit) {
  if (!c || (n && !keys) || (key_bits != 32 && key_bits != 64)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  const size_t kb = key_bits / 8;
  void *ka = nullptr, *kbuf = nullptr;
  uint32_t *va = nullptr, *vb = nullptr;
  HIPCHK(c, hipMalloc(&ka, n * kb));
  HIPCHK(c, hipMalloc(&kbuf, n * kb));
  HIPCHK(c, hipMemcpy(ka, keys, n * kb, hipMemcpyHostToDevice));
  if (vals) {
    HIPCHK(c, hipMalloc((void**)&va, n * 4));
    HIPCHK(c, hipMalloc((void**)&vb, n * 4));
    HIPCHK(c, hipMemcpy(va, vals, n * 4, hipMemcpyHostToDevice));
  }
  void* kres = nullptr;
  uint32_t* vres = nullptr;
  int rc = KS_OK;
  if (key_bits == 32) {
    uint32_t* r = nullptr;
    rc = vals ? sort_pairs(c, (uint32_t*)ka, (uint32_t*)kbuf, va, vb, n, end_bit, &r, &vres)
              : sort_keys(c, (uint32_t*)ka, (uint32_t*)kbuf, n, end_bit, &r);
    kres = r;
  } else {
    uint64_t* r = nullptr;
    rc = vals ? sort_pairs(c, (uint64_t*)ka, (uint64_t*)kbuf, va, vb, n, end_bit, &r, &vres)
              : sort_keys(c, (uint64_t*)ka, (uint64_t*)kbuf, n, end_bit, &r);
    kres = r;
  }
  if (rc == KS_OK) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(keys, kres, n * kb, hipMemcpyDeviceToHost));
    if (vals) HIPCHK(c, hipMemcpy(vals, vres, n * 4, hipMemcpyDeviceToHost));
  }
  (void)hipFree(ka);
  (void)hipFree(kbuf);
  if (va) (void)hipFree(va);
  if (vb) (void)hipFree(vb);
  return rc;
}

int ks_get_tile_keys(ks_ctx* c, uint64_t* out, size_t cap, size_t* n) {
  if (!c || !n) return KS_ERR_INVALID_ARG;
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n = c->tiles_initialised;
  const size_t m = std::min<size_t>(cap, *n);
  if (out && m) HIPCHK(c, hipMemcpy(out, c->table.slot_keys, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return KS_OK;
}

int ks_export_tiles_device(ks_ctx* c, const uint32_t* slots, size_t n, void* d_payload) {
  if (!c || (n && (!slots || !d_payload))) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (int rc = quiesce(c)) return rc;
  for (size_t i = 0; i < n; ++i)
    if (slots[i] >= c->tiles_initialised) return KS_ERR_INVALID_ARG;
  if (int rc = ensure_exchange(c, n)) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_xchg_u32, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_export_tiles, dim3((uint32_t)n), dim3(512), 0, c->stream, c->pool, c->d_xchg_u32, (uint4*)d_payload);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KS_OK;
}

int ks_merge_tiles_device(ks_ctx* c, const uint64_t* keys, size_t n, const void* d_payload) {
  if (!c || (n && (!keys || !d_payload))) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (c->fatal) return KS_ERR_INVALID_ARG;
  // group the incoming tiles by key, keeping the caller's order inside a group
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
  std::vector<uint64_t> ukeys;
  std::vector<uint32_t> offs;
  for (size_t i = 0; i < n; ++i) {
    if (i == 0 || keys[order[i]] != keys[order[i - 1]]) {
      ukeys.push_back(keys[order[i]]);
      offs.push_back((uint32_t)i);
    }
  }
  offs.push_back((uint32_t)n);
  const size_t nu = ukeys.size();
  int rc;
  if ((rc = quiesce(c))) return rc;
  if ((rc = ensure_exchange(c, n + 1))) return rc;
  uint32_t* d_offs = c->d_xchg_u32;
  uint32_t* d_idx = c->d_xchg_u32 + (nu + 1);
  HIPCHK(c, hipMemcpyAsync(c->d_xchg_u64, ukeys.data(), nu * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_offs, offs.data(), (nu + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_idx, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  // allocate tiles this rank has not seen yet
  if ((rc = insert_tiles(c, c->d_xchg_u64, nu))) return rc;
  switch (c->cfg.color_mode) {
    case KS_COLOR_MODE_COLOR:
      hipLaunchKernelGGL(k_merge_tiles<KS_COLOR_MODE_COLOR>, dim3((uint32_t)nu), dim3(512), 0, c->stream, c->table, c->pool,
                         c->d_xchg_u64, d_offs, d_idx, (const uint4*)d_payload, c->cfg.max_weight, c->d_label_lut);
      break;
    case KS_COLOR_MODE_SEMANTIC:
      hipLaunchKernelGGL(k_merge_tiles<KS_COLOR_MODE_SEMANTIC>, dim3((uint32_t)nu), dim3(512), 0, c->stream, c->table,
                         c->pool, c->d_xchg_u64, d_offs, d_idx, (const uint4*)d_payload, c->cfg.max_weight, c->d_label_lut);
      break;
    default:
      hipLaunchKernelGGL(k_merge_tiles<KS_COLOR_MODE_SEMANTIC_PROBABILITY>, dim3((uint32_t)nu), dim3(512), 0, c->stream,
                         c->table, c->pool, c->d_xchg_u64, d_offs, d_idx, (const uint4*)d_payload, c->cfg.max_weight,
                         c->d_label_lut);
      break;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return KS_OK;
}

int ks_reset_tiles(ks_ctx* c, const uint32_t* slots, size_t n) {
  if (!c || (n && !slots)) return KS_ERR_INVALID_ARG;
  if (n == 0) return KS_OK;
  if (int rc = quiesce(c)) return rc;
  for (size_t i = 0; i < n; ++i)
    if (slots[i] >= c->tiles_initialised) return KS_ERR_INVALID_ARG;
  if (int rc = ensure_exchange(c, n)) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_xchg_u32, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_reset_tiles, dim3((uint32_t)n), dim3(512), 0, c->stream, c->pool, c->d_xchg_u32);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KS_OK;
}

// ---- ks_reduce: the frame-sharded path's one exchange step, through RCCL ------------------
// librccl is loaded on first use (the library itself has no link-time dependency on it); the communicator
// is the caller's.  KS_RCCL_LIB overrides the path.
namespace {
struct RcclApi {
  void* h = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) err_string = nullptr;
  bool load(std::string* why) {
    if (h) return true;
    const char* env = getenv("KS_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      if (!nm) continue;
      h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) { *why = "librccl not found (set KS_RCCL_LIB)"; return false; }
    all_gather = (decltype(all_gather))dlsym(h, "ncclAllGather");
    send = (decltype(send))dlsym(h, "ncclSend");
    recv = (decltype(recv))dlsym(h, "ncclRecv");
    group_start = (decltype(group_start))dlsym(h, "ncclGroupStart");
    group_end = (decltype(group_end))dlsym(h, "ncclGroupEnd");
    err_string = (decltype(err_string))dlsym(h, "ncclGetErrorString");
    if (!all_gather || !send || !recv || !group_start || !group_end) { *why = "librccl lacks a required symbol"; h = nullptr; return false; }
    return true;
  }
};
RcclApi g_rccl;

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
}  // namespace

#define NCCLCHK(ctx, expr)                                                                                    \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != ncclSuccess) {                                                                                  \
      (ctx)->err = std::string(#expr) + ": " + (g_rccl.err_string ? g_rccl.err_string(r_) : "rccl error");    \
      return KS_ERR_HIP;                                                                                      \
    }                                                                                                         \
  } while (0)

int ks_tile_owner(uint64_t tile_key, int world) { return world > 0 ? (int)(splitmix64(tile_key) % (uint64_t)world) : 0; }

// grow-only scratch of ks_reduce (no allocation in the steady state)
static int ensure_reduce_scratch(ks_ctx* c, size_t n_send, size_t n_recv, int world) {
  int rc;
  if ((size_t)world > c->rx_world) {
    if ((rc = dev_alloc(c, &c->d_rx_counts, (size_t)(world + 3) * world))) return rc;  // [world] own | [world x world] all | [world] offsets | [world] cursors
    c->rx_world = (size_t)world;
  }
  if (n_send > c->cap_tx) {
    const size_t cap = std::max<size_t>(n_send + n_send / 2, 64);
    if ((rc = dev_alloc(c, &c->d_tx_keys, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_tx_slots, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_tx_payload, cap * (size_t)KS_TILE_BYTES))) return rc;
    c->cap_tx = cap;
  }
  if (n_recv > c->cap_rx) {
    const size_t cap = std::max<size_t>(n_recv + n_recv / 2, 64);
    if ((rc = dev_alloc(c, &c->d_rx_keys, cap))) return rc;
    if ((rc = dev_alloc(c, &c->d_rx_payload, cap * (size_t)KS_TILE_BYTES))) return rc;
    c->cap_rx = cap;
  }
  return KS_OK;
}

// COLLECTIVE: every rank of the communicator must call it (a rank that returns early on a local error leaves its
// peers waiting in the exchange, as with any RCCL collective).
int ks_reduce(ks_ctx* c, void* rccl_comm, int rank, int world, ks_reduce_stats* stats) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !rccl_comm)) return KS_ERR_INVALID_ARG;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (c->fatal) return KS_ERR_INVALID_ARG;
  int rc;
  if ((rc = quiesce(c))) return rc;
  const uint32_t nt = c->tiles_initialised;
  if (stats) stats->tiles_local = nt;
  hipStream_t st = c->stream;
  if (world == 1) {  // everything is owned here: nothing travels
    if (nt) HIPCHK(c, hipMemsetAsync(c->pool.dirty, 0, nt, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return KS_OK;
  }
  std::string why;
  if (!g_rccl.load(&why)) {
    c->err = why;
    return KS_ERR_UNSUPPORTED;
  }
  ncclComm_t comm = (ncclComm_t)rccl_comm;
  if ((rc = ensure_reduce_scratch(c, 0, 0, world))) return rc;
  // 1) what goes where, counted on the device: tiles touched since the last reduce that another rank owns;
  //    every rank learns every rank's counts (world x world int32) in the same breath
  int32_t* d_own = c->d_rx_counts;
  int32_t* d_all = d_own + world;
  uint32_t* d_offs = (uint32_t*)(d_all + (size_t)world * world);
  uint32_t* d_cursor = d_offs + world;
  HIPCHK(c, hipMemsetAsync(d_own, 0, (size_t)(world + 3) * world * sizeof(int32_t), st));
  const uint32_t nb = (nt + 255) / 256;
  if (nt)
    hipLaunchKernelGGL(k_dirty_by_owner, dim3(nb), dim3(256), 0, st, c->pool, (const uint64_t*)c->table.slot_keys, nt, (uint32_t)rank,
                       (uint32_t)world, 0, d_own, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t*)nullptr);
  NCCLCHK(c, g_rccl.all_gather(d_own, d_all, (size_t)world, ncclInt32, comm, st));
  std::vector<int32_t> all_counts((size_t)world * world);
  HIPCHK(c, hipMemcpyAsync(all_counts.data(), d_all, all_counts.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  std::vector<size_t> send_counts(world), recv_counts(world), recv_off(world + 1, 0), send_off(world + 1, 0);
  for (int p = 0; p < world; ++p) {
    send_counts[p] = (size_t)all_counts[(size_t)rank * world + p];
    recv_counts[p] = (size_t)all_counts[(size_t)p * world + rank];
    send_off[p + 1] = send_off[p] + send_counts[p];
    recv_off[p + 1] = recv_off[p] + recv_counts[p];
  }
  const size_t n_send = send_off[world], n_recv = recv_off[world];
  if ((rc = ensure_reduce_scratch(c, n_send, n_recv, world))) return rc;
  // 2) the send list (slots + keys grouped by owner) and the raw tile records, all on the device
  if (n_send) {
    std::vector<uint32_t> offs32(world);
    for (int p = 0; p < world; ++p) offs32[p] = (uint32_t)send_off[p];
    HIPCHK(c, hipMemcpyAsync(d_offs, offs32.data(), world * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_dirty_by_owner, dim3(nb), dim3(256), 0, st, c->pool, (const uint64_t*)c->table.slot_keys, nt, (uint32_t)rank,
                       (uint32_t)world, 1, d_own, (const uint32_t*)d_offs, d_cursor, c->d_tx_slots, c->d_tx_keys);
    hipLaunchKernelGGL(k_export_tiles, dim3((uint32_t)n_send), dim3(512), 0, st, c->pool, (const uint32_t*)c->d_tx_slots,
                       (uint4*)c->d_tx_payload);
    HIPCHK(c, hipStreamSynchronize(st));  // (offs32 is a stack-side buffer)
  }
  // 3) keys and raw tile records: one grouped exchange — on a fully connected xGMI node a rank talks to all its
  //    peers at once (a ring all-reduce would be per-link bound and move every tile through every rank)
  NCCLCHK(c, g_rccl.group_start());
  for (int peer = 0; peer < world; ++peer) {
    if (peer == rank) continue;
    if (send_counts[peer]) {
      NCCLCHK(c, g_rccl.send(c->d_tx_keys + send_off[peer], send_counts[peer], ncclUint64, peer, comm, st));
      NCCLCHK(c, g_rccl.send(c->d_tx_payload + send_off[peer] * (size_t)KS_TILE_BYTES, send_counts[peer] * (size_t)KS_TILE_BYTES,
                             ncclUint8, peer, comm, st));
    }
    if (recv_counts[peer]) {
      NCCLCHK(c, g_rccl.recv(c->d_rx_keys + recv_off[peer], recv_counts[peer], ncclUint64, peer, comm, st));
      NCCLCHK(c, g_rccl.recv(c->d_rx_payload + recv_off[peer] * (size_t)KS_TILE_BYTES, recv_counts[peer] * (size_t)KS_TILE_BYTES,
                             ncclUint8, peer, comm, st));
    }
  }
  NCCLCHK(c, g_rccl.group_end());
  std::vector<uint64_t> k_host(n_recv);
  if (n_recv) HIPCHK(c, hipMemcpyAsync(k_host.data(), c->d_rx_keys, n_recv * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  // 4) the owner folds what it received into its map, tiles of one key in ascending source-rank order
  //    (the receive buffer is ordered by source rank), in one launch
  if (n_recv && (rc = ks_merge_tiles_device(c, k_host.data(), n_recv, c->d_rx_payload))) return rc;
  // 5) what was sent starts over as an empty delta here: a later reduce cannot count it twice
  if (n_send) hipLaunchKernelGGL(k_reset_tiles, dim3((uint32_t)n_send), dim3(512), 0, st, c->pool, (const uint32_t*)c->d_tx_slots);
  HIPCHK(c, hipMemsetAsync(c->pool.dirty, 0, c->tiles_initialised, st));  // owned tiles: authoritative here, nothing pending
  HIPCHK(c, hipStreamSynchronize(st));
  if (stats) {
    stats->tiles_sent = n_send;
    stats->tiles_received = n_recv;
    stats->bytes_sent = n_send * (uint64_t)(KS_TILE_BYTES + 8);
  }
  return KS_OK;
}

// Owner: one frame's records (of the tiles this rank owns, in integration order) into the map.
static int shard_apply_segment(ks_ctx* o, const uint64_t* d_gkey, const uint32_t* d_seq, const float* d_sdf, const float* d_uw, size_t n,
                               size_t tiles_at_most) {
  if (n == 0) return KS_OK;
  if (n >= (size_t)1 << 31) { o->err = "ks_integrate_round_exact: more than 2^31 updates of one frame for one owner"; return KS_ERR_INVALID_ARG; }
  int rc;
  if (n > o->cap_sh_rx) {
    const size_t cap = std::max<size_t>(n + n / 4, 1 << 18);
    if ((rc = dev_alloc(o, &o->d_sh_tk, cap))) return rc;
    for (int b = 0; b < 2; ++b) {
      if ((rc = dev_alloc(o, &o->d_sh_pairs[b], cap))) return rc;
      if ((rc = dev_alloc(o, &o->d_sh_vals[b], cap))) return rc;
    }
    o->cap_sh_rx = cap;
  }
  hipStream_t st = o->stream;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(k_shard_tile_keys, dim3(nb), dim3(256), 0, st, (uint32_t)n, d_gkey, o->d_sh_tk);
  // get-or-insert + initialisation of the new tiles (the pool grows if it must; the records cannot name more tiles than the rank
  // that marched the frame has ever numbered)
  if ((rc = insert_tiles(o, o->d_sh_tk, n, tiles_at_most))) return rc;
  hipLaunchKernelGGL(k_shard_import, dim3(nb), dim3(256), 0, st, (uint32_t)n, o->table, o->pool, d_gkey, d_seq, o->d_sh_pairs[0], o->d_sh_vals[0]);
  const unsigned end_bit = kShardSeqBits + 9 + bits_for(o->tiles_initialised);
  uint64_t* kres = nullptr;
  uint32_t* vres = nullptr;
  // stable: a voxel's updates stay in the order they were emitted in = the integration order
  HIPCHK(o, (ksrs::sort<uint64_t, true>(o->sort_ws, o->d_sh_pairs[0], o->d_sh_pairs[1], o->d_sh_vals[0], o->d_sh_vals[1], n, std::min(56u, end_bit), st,
                                        &kres, &vres, kShardSeqBits)));
  FrameParams F{};
  const ks_config& cfg = o->cfg;
  F.log_match = o->log_match;
  F.log_non_match = o->log_non_match;
  F.tsdf.voxel_size = cfg.voxel_size;
  F.tsdf.trunc = cfg.truncation_distance;
  F.tsdf.max_weight = cfg.max_weight;
  F.tsdf.dropoff_denominator = cfg.truncation_distance - cfg.voxel_size;
  F.tsdf.sparsity_factor = cfg.sparsity_compensation_factor;
  F.tsdf.use_dropoff = cfg.use_weight_dropoff;
  F.tsdf.use_sparsity = cfg.use_sparsity_compensation_factor;
  if (cfg.color_mode == KS_COLOR_MODE_SEMANTIC)
    hipLaunchKernelGGL(k_shard_apply<KS_COLOR_MODE_SEMANTIC>, dim3(nb), dim3(256), 0, st, F, (uint32_t)n, (const uint64_t*)kres, (const uint32_t*)vres, d_sdf,
                       d_uw, o->pool, (const uint32_t*)o->d_label_lut);
  else
    hipLaunchKernelGGL(k_shard_apply<KS_COLOR_MODE_SEMANTIC_PROBABILITY>, dim3(nb), dim3(256), 0, st, F, (uint32_t)n, (const uint64_t*)kres,
                       (const uint32_t*)vres, d_sdf, d_uw, o->pool, (const uint32_t*)o->d_label_lut);
  HIPCHK(o, hipStreamSynchronize(st));
  HIPCHK(o, hipGetLastError());
  return KS_OK;
}

int ks_integrate_round_exact(ks_ctx* m, ks_ctx* o, void* rccl_comm, int rank, int world, uint64_t first_frame, const float T[7],
                             const float* xyz, const uint8_t* rgba, const uint8_t* labels, size_t n, int freespace, ks_round_stats* stats) {
  if (!m || !o || m == o || !T || world < 1 || world > 64 || rank < 0 || rank >= world || (world > 1 && !rccl_comm) || (n && !xyz))
    return KS_ERR_INVALID_ARG;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (m->fatal || o->fatal) return KS_ERR_INVALID_ARG;
  for (ks_ctx* c : {m, o}) {
    const ks_config& k = c->cfg;
    const bool frames_independent = !c->uses_early_out || k.clear_checks_every_n_frames <= 1;
    if (k.method != KS_METHOD_FAST || k.color_mode == KS_COLOR_MODE_COLOR || k.pipeline_frames != 0 || !frames_independent ||
        k.integration_order_mode == KS_ORDER_SORTED) {
      c->err = "ks_integrate_round_exact: `fast`, colours from the labels, one frame at a time, mixed order, clear_checks_every_n_frames = 1 "
               "(anything else: integrate per rank and ks_reduce)";
      return KS_ERR_UNSUPPORTED;
    }
  }
  int rc;
  const uint64_t my_frame = first_frame + (uint64_t)rank;
  if (m->shard_frames_seen > my_frame) {
    m->err = "ks_integrate_round_exact: rounds must come in frame order";
    return KS_ERR_INVALID_ARG;
  }
  m->shard_export = true;
  m->shard_world = world;
  // the frames other ranks march in between advance this marcher's set offsets and frame counters like empty clouds
  // ([K:src/semantic_tsdf_integrator_fast.cpp:165-170]: the bookkeeping is per call)
  static const uint8_t no_label = 0;
  const float T0[7] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (; m->shard_frames_seen < my_frame; ++m->shard_frames_seen)
    if ((rc = ks_integrate_points(m, T0, nullptr, nullptr, &no_label, 0, 0, nullptr))) return rc;
  std::memset(m->sh_counts, 0, sizeof(m->sh_counts));
  m->sh_exported = 0;
  ks_frame_stats fst{};
  if ((rc = ks_integrate_points(m, T, xyz, rgba, labels ? labels : (rgba ? nullptr : &no_label), n, freespace, &fst))) return rc;
  ++m->shard_frames_seen;
  // where this frame's records sit, by owner
  std::vector<size_t> send_counts(world), send_off(world + 1, 0);
  for (int p = 0; p < world; ++p) {
    send_counts[p] = m->sh_counts[p];
    send_off[p + 1] = send_off[p] + send_counts[p];
  }
  if (send_off[world] != m->sh_exported) {
    m->err = "ks_integrate_round_exact: the per-owner counts do not add up to the frame's updates";
    return KS_ERR_HIP;
  }
  uint64_t applied = 0, origin = m->sh_counts[world];
  if (world == 1) {
    if ((rc = shard_apply_segment(o, m->d_sh_gkey[1], m->d_sh_seq[1], m->d_sh_sdf[1], m->d_sh_uw[1], send_counts[0], m->tiles_initialised))) return rc;
    applied = send_counts[0];
  } else {
    std::string why;
    if (!g_rccl.load(&why)) {
      o->err = why;
      return KS_ERR_UNSUPPORTED;
    }
    ncclComm_t comm = (ncclComm_t)rccl_comm;
    hipStream_t st = o->stream;
    // 1) everybody's counts, origin-voxel flags and marchers' tile counts: (world + 2) x world
    const int W2 = world + 2;
    if ((rc = ensure_reduce_scratch(o, 0, 0, W2))) return rc;
    int32_t* d_own = o->d_rx_counts;
    int32_t* d_all = d_own + W2;
    std::vector<int32_t> own(W2);
    for (int p = 0; p < world; ++p) own[p] = (int32_t)send_counts[p];
    own[world] = (int32_t)m->sh_counts[world];
    own[world + 1] = (int32_t)m->tiles_initialised;
    HIPCHK(o, hipMemcpyAsync(d_own, own.data(), own.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    NCCLCHK(o, g_rccl.all_gather(d_own, d_all, (size_t)W2, ncclInt32, comm, st));
    std::vector<int32_t> all((size_t)W2 * world);
    HIPCHK(o, hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(o, hipStreamSynchronize(st));
    std::vector<size_t> recv_counts(world), recv_off(world + 1, 0);
    for (int p = 0; p < world; ++p) {
      recv_counts[p] = p == rank ? 0 : (size_t)all[(size_t)p * W2 + rank];
      recv_off[p + 1] = recv_off[p] + recv_counts[p];
      origin |= (uint64_t)all[(size_t)p * W2 + world];
    }
    const size_t n_recv = recv_off[world];
    // 2) receive buffers on the owner context (its own d_sh_*[0]: an owner context never exports)
    if (n_recv > o->cap_sh) {
      const size_t cap = std::max<size_t>(n_recv + n_recv / 4, 1 << 18);
      if ((rc = dev_alloc(o, &o->d_sh_gkey[0], cap))) return rc;
      if ((rc = dev_alloc(o, &o->d_sh_seq[0], cap))) return rc;
      if ((rc = dev_alloc(o, &o->d_sh_sdf[0], cap))) return rc;
      if ((rc = dev_alloc(o, &o->d_sh_uw[0], cap))) return rc;
      o->cap_sh = cap;
    }
    // 3) one grouped exchange: every rank talks to all its peers at once (xGMI is point to point)
    NCCLCHK(o, g_rccl.group_start());
    for (int peer = 0; peer < world; ++peer) {
      if (peer == rank) continue;
      if (send_counts[peer]) {
        NCCLCHK(o, g_rccl.send(m->d_sh_gkey[1] + send_off[peer], send_counts[peer], ncclUint64, peer, comm, st));
        NCCLCHK(o, g_rccl.send(m->d_sh_seq[1] + send_off[peer], send_counts[peer], ncclUint32, peer, comm, st));
        NCCLCHK(o, g_rccl.send(m->d_sh_sdf[1] + send_off[peer], send_counts[peer], ncclFloat32, peer, comm, st));
        NCCLCHK(o, g_rccl.send(m->d_sh_uw[1] + send_off[peer], send_counts[peer], ncclFloat32, peer, comm, st));
      }
      if (recv_counts[peer]) {
        NCCLCHK(o, g_rccl.recv(o->d_sh_gkey[0] + recv_off[peer], recv_counts[peer], ncclUint64, peer, comm, st));
        NCCLCHK(o, g_rccl.recv(o->d_sh_seq[0] + recv_off[peer], recv_counts[peer], ncclUint32, peer, comm, st));
        NCCLCHK(o, g_rccl.recv(o->d_sh_sdf[0] + recv_off[peer], recv_counts[peer], ncclFloat32, peer, comm, st));
        NCCLCHK(o, g_rccl.recv(o->d_sh_uw[0] + recv_off[peer], recv_counts[peer], ncclFloat32, peer, comm, st));
      }
    }
    NCCLCHK(o, g_rccl.group_end());
    HIPCHK(o, hipStreamSynchronize(st));
    // 4) the frames of the round in frame order = in the order of the ranks that marched them
    for (int src = 0; src < world; ++src) {
      if (src == rank) {
        if ((rc = shard_apply_segment(o, m->d_sh_gkey[1] + send_off[rank], m->d_sh_seq[1] + send_off[rank], m->d_sh_sdf[1] + send_off[rank],
                                      m->d_sh_uw[1] + send_off[rank], send_counts[rank], m->tiles_initialised)))
          return rc;
        applied += send_counts[rank];
      } else {
        if ((rc = shard_apply_segment(o, o->d_sh_gkey[0] + recv_off[src], o->d_sh_seq[0] + recv_off[src], o->d_sh_sdf[0] + recv_off[src],
                                      o->d_sh_uw[0] + recv_off[src], recv_counts[src], (size_t)all[(size_t)src * W2 + world + 1])))
          return rc;
        applied += recv_counts[src];
      }
    }
    if (stats) stats->bytes_sent = (m->sh_exported - send_counts[rank]) * 20ull;
  }
  if (stats) {
    stats->updates_marched = m->sh_exported;
    stats->updates_applied = applied;
    stats->origin_voxel_touched = origin ? 1 : 0;
    stats->rays_cast = fst.n_rays_cast;
  }
  return KS_OK;
}

// keep_integrator_state: only the MAP goes (tile table, pool flags); the two approximate sets, their offsets, the
// frame counters and the early-out table stay as the frames so far left them — what vxb::TsdfServer::clear() does to the
// reference's integrator, which it does not touch.  Frames in flight are completed first (their stage B has already
// entered its marks); without it a frame that was never applied is dropped with the map.
static int clear_impl(ks_ctx* c, bool keep_integrator_state) {
  if (keep_integrator_state) {
    if (int rc = quiesce(c)) return rc;
    c->owed = ks_frame_stats{};
  }
  for (auto& S : c->slot) S.pending = false;  // a frame that was never applied is dropped with the map
  c->batch_slots.clear();
  c->owed = ks_frame_stats{};
  if (c->stream_tail != c->stream) HIPCHK(c, hipStreamSynchronize(c->stream_tail));
  if (c->stream_long) HIPCHK(c, hipStreamSynchronize(c->stream_long));  // (long runs of the last frame: deferred join)
  c->pending_join = nullptr;
  if (int rc = sync_march(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemset(c->table.ent, 0xff, ((size_t)c->table.mask + 1) * sizeof(TileEntry)));
  HIPCHK(c, hipMemset(c->pool.updated, 0, c->cfg.max_tiles));
  HIPCHK(c, hipMemset(c->pool.dirty, 0, c->cfg.max_tiles));
  HIPCHK(c, hipMemset(c->d_state, 0, 64 * (kSlots + 1)));
  if (!keep_integrator_state) {
    // a cleared context behaves like a fresh one: both approximate sets as their constructor leaves them
    HIPCHK(c, hipMemset(c->d_start_set, 0, sizeof(uint64_t) << kSetBits));
    for (int t = 0; t < c->n_obs; ++t) {
      HIPCHK(c, hipMemset(c->d_observed_[t], 0, 2 * (sizeof(uint64_t) << kSetBits)));
      HIPCHK(c, hipMemcpy(c->d_observed_[t], &kObsPoison, 8, hipMemcpyHostToDevice));
    }
    const uint64_t poison = ~0ull;
    HIPCHK(c, hipMemcpy(c->d_start_set, &poison, 8, hipMemcpyHostToDevice));
    if (c->d_eo_plain) {
      HIPCHK(c, hipMemset(c->d_eo_plain, 0, sizeof(uint64_t) << kSetBits));
      HIPCHK(c, hipMemcpy(c->d_eo_plain, &poison, 8, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemset(c->d_eo_committed, 0, 64));
      c->eo_frame_no = 0;
      c->eo_last_commit = nullptr;
    }
    c->start_offset = c->observed_offset = 0;
    c->reset_counter = 0;
    c->obs_tag = 0;
    c->obs_tag_lo = 1;
  }
  c->tiles_initialised = 0;
  for (auto& S : c->slot)
    if (S.h_snap) std::memset(S.h_snap, 0, sizeof(HostSnap));  // (the pool-growth trigger reads the snapshots' tile counts)
  c->fatal = false;
  return KS_OK;
}

int ks_clear(ks_ctx* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  return clear_impl(c, false);
}

int ks_clear_voxels(ks_ctx* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  return clear_impl(c, true);
}

int ks_flush(ks_ctx* c, ks_frame_stats* stats) {
  if (!c) return KS_ERR_INVALID_ARG;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const int rc = flush_pending(c);
  deliver_stats(c, stats);
  return rc;
}

int ks_synchronize(ks_ctx* c) {
  if (!c) return KS_ERR_INVALID_ARG;
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KS_OK;
}

void* ks_stream(ks_ctx* c) { return c ? (void*)c->stream : nullptr; }

#ifdef KS_STATS
// diagnostics build only: read (and clear) the k_test counters
int ks_debug_test_stats(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_test_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return KS_ERR_HIP;
  unsigned long long z[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_test_stats), z, sizeof(z)) != hipSuccess) return KS_ERR_HIP;
  return KS_OK;
}
#endif

int ks_early_out_iterations(ks_ctx* c, uint64_t* frames, uint64_t* iterations) {
  if (!c) return KS_ERR_INVALID_ARG;
  if (frames) *frames = c->eo_frames;
  if (iterations) *iterations = c->eo_iterations;
  return KS_OK;
}

int ks_early_out_stats(ks_ctx* c, uint64_t out[5]) {
  if (!c || !out) return KS_ERR_INVALID_ARG;
  out[0] = c->eo_frames;
  out[1] = c->eo_iterations;
  out[2] = c->eo_fallbacks.load(std::memory_order_relaxed);
  out[3] = (c->eo_device && !c->eo_device_off) ? 1 : 0;
  out[4] = (c->exact_early_out && c->cfg.pipeline_frames > 0) ? 1 : 0;
  return KS_OK;
}

int ks_update_stats(ks_ctx* c, uint64_t out[4]) {
  if (!c || !out) return KS_ERR_INVALID_ARG;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!c->d_xl_hdr) return KS_OK;
  if (int rc = quiesce(c)) return rc;
  if (c->stream_xlong) HIPCHK(c, hipStreamSynchronize(c->stream_xlong));
  XlHeader h;
  HIPCHK(c, hipMemcpy(&h, c->d_xl_hdr, sizeof(h), hipMemcpyDeviceToHost));
  out[0] = h.tot_walked;
  out[1] = h.tot_fallback;
  out[2] = h.tot_chunks;
  out[3] = h.tot_replayed;
  return KS_OK;
}

int ks_pipeline_shape(ks_ctx* c, int32_t out[4]) {
  if (!c || !out) return KS_ERR_INVALID_ARG;
  out[0] = c->cfg.pipeline_frames;
  out[1] = c->n_slots;
  out[2] = c->batch;
  out[3] = c->n_march;
  return KS_OK;
}

int ks_profile_enable(ks_ctx* c, int level) {
  if (!c || level < 0 || level > 2) return KS_ERR_INVALID_ARG;
  c->profiling = level;
  return KS_OK;
}

int ks_profile_get(ks_ctx* c, ks_profile* out, int reset) {
  if (!c || !out) return KS_ERR_INVALID_ARG;
  if (int rc = quiesce(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < kProfSets; ++i) resolve_prof(c, i);
  *out = c->prof;
  if (reset) c->prof = ks_profile{};
  return KS_OK;
}

}  // extern "C"
