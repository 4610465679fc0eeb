// pllgpu.hip - device context, data movement and kernel launches behind include/pll_amd_device.h.
// gfx950 (MI355X) only. No CPU path lives here: every compute entry point launches HIP kernels
// or fails with an error.
#include "../../../include/pll_amd_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <vector>

#include "kernels_common.h"
#include "kernels_dna.h"
#include "kernels_generic.h"
#include "kernels_deriv.h"
#include "kernels_mfma.h"
#include "kernels_mfma_wide.h"
#include "kernels_lean.h"
#include "kernels_repeats.h"

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(call)                                                                            \
  do                                                                                             \
  {                                                                                              \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(e_ == hipErrorOutOfMemory ? PLLGPU_ENOMEM : PLLGPU_ERUNTIME, "%s failed: %s",   \
                  #call, hipGetErrorString(e_));                                                 \
  } while (0)

// Cached launch plans hold raw device pointers: whenever one of a context's blocks is (re)allocated or freed, THAT
// context's epoch moves (pllgpu_ctx::alloc_epoch) and its plans are planned again. Round 5 kept one counter for the
// whole process, so a partition created, grown or destroyed anywhere - the usual shape of a multi-partition caller, or a
// flat pll_core_* call beside a partition - dropped every other partition's plans. A DevBuf does not know its owner:
// the entry point that is running says whose blocks are being touched (DeviceScope sets t_epoch; a context is used by
// one thread at a time, so the counter is plain).
static unsigned long long g_orphan_epoch = 1; // blocks touched outside any entry point (none today)
static thread_local unsigned long long *t_epoch = &g_orphan_epoch;

// Device blocks a context has let go of are kept for its next request instead of going back to the runtime: hipFree
// waits for the whole device (35 us on average, 0.5 ms at worst, measured under a tree search on a site-repeats partition
// - every move re-orients nodes whose class counts, and so their CLV / map buffers, change size: 9 pairs of hipFree +
// hipMalloc per move, 330 us of a 450 us move; profiles/r6_tree_search.json). Everything a context does is ordered by its
// one stream, so a block may serve its next owner at once: whatever was enqueued to read the old contents runs before
// whatever is enqueued to write the new ones. Sizes are rounded up to four significant bits (at most 1/8 more), so that
// blocks of nearly the same size are interchangeable; what is held idle is bounded (a quarter of what the context's
// buffers hold, and a failed hipMalloc gives everything back and tries again); pllgpu_destroy returns it all.
struct BlockPool
{
  std::multimap<size_t, void *> idle; // bytes -> block
  size_t idle_bytes = 0;
  size_t live_bytes = 0;              // blocks the context's buffers hold
  void trim(size_t keep)
  {
    while (idle_bytes > keep && !idle.empty())
    {
      auto it = std::prev(idle.end()); // the largest first
      (void)hipFree(it->second);
      idle_bytes -= it->first;
      idle.erase(it);
    }
  }
};
// idle blocks a context may hold: a quarter of what its buffers hold (at least 64 MB, at most 8 GB) - many partitions of a
// multi-partition caller must not each sit on gigabytes they once needed
constexpr size_t kPoolIdleMax = (size_t)8 << 30, kPoolIdleMin = (size_t)64 << 20;
static thread_local BlockPool *t_pool = nullptr; // the running entry point's context (DeviceScope), like t_epoch

static inline size_t block_bytes(size_t bytes)
{
  if (bytes <= 4096) return 4096;
  int top = 63 - __builtin_clzll((unsigned long long)bytes);
  const size_t step = (size_t)1 << (top - 3);
  return (bytes + step - 1) & ~(step - 1);
}

static hipError_t block_take(size_t bytes, void **out)
{
  if (t_pool)
  {
    auto it = t_pool->idle.find(bytes);
    if (it != t_pool->idle.end())
    {
      *out = it->second;
      t_pool->idle_bytes -= it->first;
      t_pool->live_bytes += bytes;
      t_pool->idle.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && t_pool && !t_pool->idle.empty())
  {
    (void)hipGetLastError();
    t_pool->trim(0);
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess && t_pool) t_pool->live_bytes += bytes;
  return e;
}

static void block_give(void *p, size_t bytes)
{
  if (!t_pool)
  {
    (void)hipFree(p);
    return;
  }
  t_pool->idle.emplace(bytes, p);
  t_pool->idle_bytes += bytes;
  t_pool->live_bytes -= std::min(bytes, t_pool->live_bytes);
  const size_t cap = std::min(kPoolIdleMax, std::max(kPoolIdleMin, t_pool->live_bytes / 4u));
  if (t_pool->idle_bytes > cap) t_pool->trim(cap / 2u);
}

template <typename T>
struct DevBuf
{
  T *p = nullptr;
  size_t cap = 0; // elements
  int ensure(size_t n)
  {
    if (n <= cap) return 0;
    ++*t_epoch;
    if (p)
    {
      block_give(p, cap * sizeof(T)); // (its readers and writers so far are ahead of the next owner's on the stream)
      p = nullptr;
      cap = 0;
    }
    const size_t bytes = block_bytes(n * sizeof(T));
    void *q = nullptr;
    HIP_TRY(block_take(bytes, &q));
    p = static_cast<T *>(q);
    cap = bytes / sizeof(T);
    return 0;
  }
  void release()
  {
    if (p)
    {
      block_give(p, cap * sizeof(T));
      ++*t_epoch;
    }
    p = nullptr;
    cap = 0;
  }
};

// mapped result block: [0..7] lnL / derivative words, [8..) ascertainment-bias terms (up to 64 x 4 doubles)
constexpr size_t kResultBytes = (8 + 4 * 64) * sizeof(double);
constexpr unsigned kAscOff = 8;
constexpr unsigned kRepHostCap = 1u << 16; // ops of one class-map call whose counts the mapped block holds (larger calls go in pieces)
constexpr unsigned char kMap8 = 1, kMap32 = 2;

struct pllgpu_ctx
{
  unsigned long long alloc_epoch = 1; // moves whenever one of this context's device blocks is (re)allocated or freed
  BlockPool pool;                     // device blocks this context has let go of (DevBuf)
  pllgpu_geometry_t geo;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  // derived kernel geometry
  GenGeo gg;
  int ich = 0;
  bool dna_fast = false;
  bool use_mfma = false;
  int mfma_ng = 16;         // 4-state groups the MFMA kernels run with: 16 (33..64 states), 8 (21..32), 5 (17..20)
  unsigned mfma_wide = 1;   // 33..64 states, inner x inner: k_partials_mfma_wide (kernels_mfma_wide.h); PLL_AMD_MFMA_WIDE=0 (A/B, parity tests): k_partials_mfma
  bool mfma_pad = false;    // PLL_AMD_MFMA_PAD=1 (A/B): 61 states through the padded 64-state contraction
  unsigned xcd_order = 1;   // PLL_AMD_NO_XCD_ORDER=1 (A/B): store-bound launches in the natural workgroup order (kernels_common.h: xcd_block)
  bool tiled = false;       // generic shapes keep CLVs in the tiled sites-contiguous layout
  DevBuf<double> scratch;   // host-layout staging for mirror copies of tiled CLVs
  size_t pm_stride = 0; // doubles per matrix in PT layout
  unsigned span = 0;

  std::vector<DevBuf<double>> clv;
  std::vector<unsigned char> clv_aos;   // 4x4: the node's device CLV is entry-contiguous (class-compressed node)
  std::vector<DevBuf<unsigned>> scaler;
  std::vector<DevBuf<unsigned char>> tipchars;
  std::vector<DevBuf<unsigned>> site_id, id_site;
  std::vector<DevBuf<unsigned char>> site_id8; // site -> class in bytes: nodes with <= kRepNarrow classes (kernels_repeats.h)
  std::vector<unsigned char> map_forms;        // per node: which forms of its site -> class map are current (kMap8 | kMap32)
  std::vector<DevBuf<unsigned>> lent, rent; // class -> child entry, per compressed node (kernels_repeats.h)
  std::vector<int> rep_left, rep_right;     // the children those maps were built for, or -1
  std::vector<unsigned> ids;
  DevBuf<unsigned long long> tipmap;
  bool tipmap_set = false;
  DevBuf<double> pmat, freqs, rate_weights, prop_invar, persite, block_sums;
  DevBuf<double> edge_partials;   // k_edge_mfma: [rate][row group][site] partial site likelihoods
  DevBuf<unsigned> edge_tickets;  // ... and the ticket per item block (zero between launches)
  DevBuf<unsigned> counter;
  DevBuf<unsigned char> mfma_flags;      // [op in launch][rate][entry] scaling decisions (kernels_mfma.h)
  DevBuf<double> eigenvals, rates, diag; // derivatives: [rate_matrices][SP], [R], [R][S][4]
  DevBuf<double> evecs, ievecs, brlen;   // device P-matrices: [rate_matrices][S][SP] x 2, staged branch lengths
  DevBuf<unsigned> mindex;               // staged matrix indices
  DevBuf<unsigned> rep_table, rep_blocksum, rep_counts; // site-repeats class computation (kernels_repeats.h): arena, bitmaps, counts per op of a call
  DevBuf<unsigned> rep_sync;             // [kRepOps] tickets per op of a launch, then the launch's own (zero between launches)
  DevBuf<unsigned char> rep_ops;         // the op descriptors of one call
  std::vector<RepOp> rep_ops_host;
  unsigned rep_wgs = 0;                  // PLL_AMD_REP_WGS: workgroups per op of k_rep_mark (0: by the launch's size)
  unsigned rep_max_ranges = 8;           // PLL_AMD_REP_RANGES: site ranges per part of a large table, at most (kernels_repeats.h)
  bool sub_pack_always = false;          // PLL_AMD_SUB_PACK_ALWAYS=1: k_sub_pack after every class-map call, whatever it reported (A/B)
  bool rep_bits = true;                  // PLL_AMD_REP_BITS=0: the bitmap of first sites by atomics + k_rep_scan for every table size (A/B)
  bool rep_hints = true;                 // PLL_AMD_REP_HINTS=0: every level of a class-map call is launched (A/B, tests)
  unsigned long long rep_hint_key = 0;   // the last call: a hash of its (parent, left, right, level) records ...
  unsigned rep_hint_count = 0, rep_hint_level = 0; // ... their number, the highest level with a compressed parent ...
  bool rep_hint_any = false;                       // ... if there was one
  bool rep_scratch_dirty = false;        // a class-map call did not complete: its bitmaps and tickets are cleared before the next
  DevBuf<double> sumtable[PLLGPU_SUMTABLE_SLOTS]; // device-resident sumtables (tiled like a CLV), allocated on first use
  double *result_dev = nullptr;  // device alias of result_host
  DevBuf<unsigned> pattern_weights;
  DevBuf<int> invariant;
  bool invariant_set = false;
  unsigned *rep_host = nullptr;  // pinned + mapped: class counts of a class-map call [rep_host_cap], the sequence word, an error word
  unsigned *rep_host_dev = nullptr;
  unsigned rep_host_cap = 0;
  unsigned rep_seq = 0;
  double *result_host = nullptr; // pinned + mapped: [0] lnL, [1] sequence of the call that wrote it
  double seq = 0.0;
  double seq_override = 0.0;     // pllgpu_edge_t.sequence of the evaluation being issued (0: number it here)
  DevBuf<double> reduce;         // {lnL, sequence}: the operand of a caller's all-reduce (pllgpu_reduce_buffer)
  std::vector<double> stage;     // host staging for the P-matrix re-layout
  std::vector<unsigned char> stage8; // ... and for a class map that goes up as bytes
  unsigned last_launches = 0;
  unsigned long long rep_ops_total = 0, rep_launches_total = 0; // class-map ops handed to the device / class kernels launched, ever
  double last_bytes = 0.0;       // algorithmic HBM bytes of the last update_partials call
  bool no_tip_columns = false;   // PLL_AMD_NO_TIP_COLUMNS=1: tips always through the FMA contraction
  bool tt_stream = true;         // PLL_AMD_TT_STREAM=0: plain tip x tip levels of 33..64 states through k_partials_mfma's column route (round 5) - the control
  bool no_par_lds = false;       // PLL_AMD_NO_PARENT_LDS=1 (A/B): entry-contiguous parents stored 8 bytes per lane
  bool no_coop_fetch = false;    // PLL_AMD_NO_COOP_FETCH=1 (A/B): the FMA kernels' entry-contiguous children fetched per lane
  size_t stream_parent_bytes = (size_t)256 << 20; // parents of a grouped launch beyond this leave with streaming stores (the Infinity Cache)
  bool defer_tail = false;       // DNA: hold the traversal's last ops for one call (k_edge_dna_tail)
  std::vector<pllgpu_op_t> deferred; // ops accepted by pllgpu_update_partials and not launched yet
  bool fuse_cc = false;          // DNA: also two producer levels under a group parent (cherry-cherry children)
  int fuse_cc16 = -1;            // DNA, chain plans: also a parent over two complete 8-tip groups (fifteen ops, k_partials_dna_cc16):
                                 // -1 by size (chain_plan.h: use_cc16), 0 never, 1 always (PLL_AMD_FUSE_CC16)
  bool fuse_gg = true;           // DNA + site repeats: groups over two gathering producers (PLL_AMD_NO_FUSE_GG=1: off)
  bool fuse = false;             // DNA: evaluate producer + consumer ops in one kernel (kernels_dna.h)
  bool chains = false;           // DNA: chain plans (k_partials_dna_chain) for dependency-only op lists
  bool any_aos = false;          // a class-compressed CLV exists (site repeats): no chain plans
  struct ChainPlan *plan = nullptr; // the last chain plan, re-launched as is when the same list comes again
  bool chain_held = false;          // the plan's last stage has not been launched yet (chain tail, kernels_dna.h)
  DevBuf<unsigned char> chain_dev;  // its descriptors
  // level-scheduled lists: the launches of a planned list are kept (descriptor packs by value) and replayed
  // as they are when the same list comes again while nothing they point at has moved (LevelPlan below)
  unsigned long long maps_epoch = 1; // bumped whenever the SHAPE of the class maps changes: a node's class count, the children its entry maps
                                     // were built for, the form of its site -> class map (cached launches carry counts and pointers)
  unsigned long long maps_version = 1; // bumped whenever class maps are written at all (data derived from their CONTENTS: k_sub_pack)
  std::vector<unsigned char> map_widened; // per node: a launch reads the 32-bit form although the class kernels write bytes (wide_map)
  std::vector<struct LevelPlan *> level_plans;
  struct LevelPlan *recording = nullptr;
  int launch_rc = 0;                // a launch helper that failed inside emit()
  int fenced = 0;                   // PLL_AMD_FENCED_HANDOFF=1 (kernels_common.h: handoff_*)
  unsigned long long plan_stamp = 0;
  unsigned long long plan_replays = 0; // op lists that were launched from a kept plan (chain or level), ever
  bool plan_cache = true;           // PLL_AMD_NO_PLAN_CACHE=1 plans every call afresh
  DevBuf<unsigned char> cherry_bits; // k_cherry_bits: which cherry entries are rescaled, [slot][rate][pair of tip codes]
  struct CherrySlot
  {
    unsigned long long lver = 0, rver = 0, maps = 0;
  };
  std::map<unsigned long long, unsigned> cherry_slot_of; // (left matrix, right matrix) -> slot
  std::vector<CherrySlot> cherry_slot;                   // what the slot's table was computed from
  std::vector<unsigned long long> pm_version;            // bumped whenever a device matrix is written
  unsigned tip_ncodes = 0;          // codes in use: 1 + the highest code with a non-empty mask
  bool generic_aos = true;          // PLL_AMD_NO_GENERIC_AOS=1: compressed nodes of non-4x4 shapes stay tiled (A/B)
  bool lean = false;                // 17..20 states, <= 4 rates: the level launches on the matrix pipe (kernels_lean.h)
  bool fuse_mfma = false;           // 17..32 states on the matrix pipe: the same groups (kernels_mfma.h: k_partials_mfma_cc)
  bool subtrees = false;            // DNA + site repeats: all-tip subtrees straight from the tip codes (subtree_plan.h)
  DevBuf<unsigned char> sub_dev;    // their descriptors on the device ...
  std::vector<SubItem> sub_cache, sub_build; // ... and what that array holds / the list being planned
  unsigned long long sub_epoch = 0;
  DevBuf<unsigned long long> sub_packed; // k_sub_pack: per sub-tree op and entry, the tip codes below it
  bool sub_pack_valid = false;           // ... formed for the descriptors on the device, the class maps (maps_version) ...
  unsigned long long sub_pack_maps = 0, sub_pack_tips = 0, tips_epoch = 1; // ... and the tip data (tips_epoch) of now
  unsigned long long maps_foreign = 1, sub_pack_foreign = 0; // maps written by anything but the class kernels (uploads, count changes)
  unsigned sub_pack_since = 0;           // the class-map calls from this sequence number on came after the packed words
  DevBuf<unsigned> rep_changed;          // RepPack::changed
  std::vector<DevBuf<unsigned>> rep_keep; // per node: RepOp::keep
  std::vector<unsigned char> rep_ops_sent; // the descriptors the device array holds (bytes), and where
  const void *rep_ops_sent_at = nullptr;
  DevBuf<unsigned> rep_final;            // RepOp::final of the deferred ops of a call (kRepFuseCells cells each)
  bool rep_fuse = true;                  // PLL_AMD_REP_FUSE=0: every op's site -> class pass in k_rep_assign (A/B)
  // small transfers go through one block of pinned, device-visible host memory (stage_take below)
  unsigned char *ring_host = nullptr, *ring_dev = nullptr;
  size_t ring_cap = 0, ring_off = 0;
  bool ring_failed = false;
  // downloads through the block that have been enqueued but not waited for (pllgpu_download_defer)
  struct PendingDown
  {
    void *host;
    const unsigned char *ring;
    size_t bytes;
  };
  std::vector<PendingDown> pending_down;
  bool defer_down = false;
};

// what pllgpu_update_partials did for one op list through the level scheduler: its launches, in order, with
// their descriptor packs captured by value; the ops it held back for the log-likelihood kernel; the CLV layout
// flags resolve_op() left behind. Valid while no device block has been (re)allocated (pllgpu_ctx::alloc_epoch) and no
// class map has changed (maps_epoch).
struct LevelPlan
{
  std::vector<pllgpu_op_t> key;
  unsigned long long alloc_epoch = 0, maps_epoch = 0;
  std::vector<std::function<void()>> launches;
  std::vector<pllgpu_op_t> deferred;
  std::vector<std::pair<unsigned, unsigned char>> aos; // (node, entry-contiguous?) as the list leaves them
  unsigned nlaunches = 0;
  double bytes = 0.0;
  bool any_aos = false;
  unsigned long long used = 0; // LRU stamp
};
constexpr size_t kLevelPlans = 4;

// run a launch now and, while a plan is being recorded, keep it
template <class F>
static inline void emit(pllgpu_ctx *c, F &&fn)
{
  fn();
  if (c->recording) c->recording->launches.emplace_back(std::forward<F>(fn));
}

static inline int use(pllgpu_ctx *c)
{
  HIP_TRY(hipSetDevice(c->device));
  return 0;
}

// Every entry point works on the context's device and leaves the calling thread's current HIP device as it
// found it (the thread may belong to torch or to another HIP user with a different device selected).
struct DeviceScope
{
  int prev = -1, rc = 0;
  unsigned long long *prev_epoch;
  BlockPool *prev_pool;
  explicit DeviceScope(pllgpu_ctx *c) : prev_epoch(t_epoch), prev_pool(t_pool)
  {
    t_epoch = &c->alloc_epoch; // whose blocks this entry point may move (entry points nest: put back on the way out)
    t_pool = &c->pool;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != c->device) rc = use(c);
    else prev = -1; // nothing to restore
  }
  ~DeviceScope()
  {
    t_epoch = prev_epoch;
    t_pool = prev_pool;
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// hipFuncSetAttribute applies to the CURRENT device: once per (kernel, device), not once per process
#include <mutex>
#include <map>
static void raise_lds_limit(const void *fn, int device, size_t dynamic_bytes)
{
  // dynamic LDS beyond the default limit has to be announced per kernel; the kernel's static LDS counts
  // against the 160 KB as well, so only what a launch needs is asked for (the largest request so far)
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> done;
  std::lock_guard<std::mutex> g(mu);
  size_t &have = done[{fn, device}];
  if (dynamic_bytes <= have || dynamic_bytes <= 32 * 1024) return;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes) != hipSuccess)
    (void)hipGetLastError(); // the launch itself will report what is wrong
  else
    have = dynamic_bytes;
}

extern "C" const char *pllgpu_last_error(void) { return g_err; }

extern "C" int pllgpu_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int pllgpu_default_device(void)
{
  const char *env = getenv("PLL_AMD_DEVICE");
  int device = 0;
  if (env && strcmp(env, "auto") == 0) return -2;
  if (env) return atoi(env);
  if (hipGetDevice(&device) != hipSuccess)
  {
    (void)hipGetLastError();
    device = 0;
  }
  return device;
}

extern "C" int pllgpu_context_device(const pllgpu_ctx_t *c) { return c ? c->device : -1; }

static void derive_geometry(pllgpu_ctx *c)
{
  const pllgpu_geometry_t &g = c->geo;
  int ich = g.states <= 4 ? 4 : g.states <= 8 ? 8 : g.states <= 16 ? 16 : g.states <= 20 ? 20 : 32;
  GenGeo &gg = c->gg;
  gg.S = g.states;
  gg.SP = g.states_padded;
  gg.R = g.rate_cats;
  gg.nchunks = (g.states + ich - 1) / ich;
  gg.SPT = gg.nchunks * ich;
  gg.tile_sz = g.rate_cats * g.states * 64u;
  gg.scale_mode = g.per_rate_scalers ? 2 : 1;
  c->ich = ich;
  c->dna_fast = (g.states == 4 && g.rate_cats == 4);
  if (const char *v = getenv("PLL_AMD_GENERIC_ONLY")) // experiment switch: route DNA through the generic kernels
    if (*v && *v != '0') c->dna_fast = false;
  c->fuse = c->dna_fast;
  if (const char *v = getenv("PLL_AMD_NO_FUSE")) // experiment switch: one kernel per op group, no producer/consumer fusion
    if (*v && *v != '0') c->fuse = false;
  if (const char *v = getenv("PLL_AMD_TT_STREAM")) c->tt_stream = !(*v == '0');
  if (const char *v = getenv("PLL_AMD_NO_TIP_COLUMNS"))
    if (*v && *v != '0') c->no_tip_columns = true;
  if (const char *v = getenv("PLL_AMD_NO_PARENT_LDS"))
    if (*v && *v != '0') c->no_par_lds = true;
  if (const char *v = getenv("PLL_AMD_NO_COOP_FETCH"))
    if (*v && *v != '0') c->no_coop_fetch = true;
  c->defer_tail = c->dna_fast;
  if (const char *v = getenv("PLL_AMD_NO_TAIL_FUSION"))
    if (*v && *v != '0') c->defer_tail = false;
  c->fuse_cc = c->fuse;
  c->fuse_cc16 = -1;
  if (const char *v16 = getenv("PLL_AMD_FUSE_CC16"))
    if (*v16) c->fuse_cc16 = *v16 != '0' ? 1 : 0;
  if (const char *v = getenv("PLL_AMD_NO_FUSE_CC"))
    if (*v && *v != '0') c->fuse_cc = false;
  c->chains = c->fuse;
  if (const char *v = getenv("PLL_AMD_NO_CHAINS"))
    if (*v && *v != '0') c->chains = false;
  if (const char *v = getenv("PLL_AMD_NO_GENERIC_AOS"))
    if (*v && *v != '0') c->generic_aos = false;
  if (const char *v = getenv("PLL_AMD_FENCED_HANDOFF"))
    if (*v && *v != '0') c->fenced = 1;
  if (const char *v = getenv("PLL_AMD_NO_PLAN_CACHE"))
    if (*v && *v != '0') c->plan_cache = false;
  if (const char *v = getenv("PLL_AMD_NO_FUSE_GG"))
    if (*v && *v != '0') c->fuse_gg = false;
  c->subtrees = c->dna_fast;
  if (const char *v = getenv("PLL_AMD_NO_SUBTREES")) // A/B switch: the bottom levels under site repeats go level by level
    if (*v && *v != '0') c->subtrees = false;
  c->tiled = true; // every shape keeps CLVs in the tiled sites-contiguous layout
  // 33..64 states: CLV updates on the fp64 matrix pipe (kernels_mfma.h); PLL_AMD_NO_MFMA=1 keeps the FMA kernel
  {
    // PLL_AMD_MFMA_MIN_STATES: from how many states on the CLV updates run on the matrix pipe
    unsigned min_states = 33;
    if (const char *v = getenv("PLL_AMD_MFMA_MIN_STATES")) min_states = (unsigned)std::max(17, atoi(v));
    c->use_mfma = (g.states >= min_states && g.rate_cats <= 16);
    c->mfma_ng = g.states > 32 ? 16 : g.states > 20 ? 8 : 5;
  }
  if (const char *v = getenv("PLL_AMD_NO_MFMA"))
    if (*v && *v != '0') c->use_mfma = false;
  if (const char *v = getenv("PLL_AMD_MFMA_WIDE")) c->mfma_wide = atoi(v) != 0 ? 1u : 0u;
  if (const char *v = getenv("PLL_AMD_MFMA_PAD")) c->mfma_pad = *v && *v != '0';
  if (const char *v = getenv("PLL_AMD_NO_XCD_ORDER")) c->xcd_order = (*v && *v != '0') ? 0u : 1u;
  // (tip x tip, tip x tip -> inner x inner) groups of 17..32 states: on the matrix pipe (kernels_mfma.h:
  // k_partials_mfma_cc), whatever pipe the level launches use - C3 (20 states) 4.0 -> 5.1 G updates/s on the same box
  // (profiles/README.md, round 2). PLL_AMD_FUSE_GENERIC=0 or PLL_AMD_NO_FUSE=1: level launches only (A/B, parity tests).
  // (The table-fed FMA groups, the 33..64-state groups and the inner x inner groups of rounds 1-2 measured slower than
  // the launches they replaced and are gone; their numbers stay in profiles/README.md.)
  bool groups = true;
  if (const char *v = getenv("PLL_AMD_FUSE_GENERIC")) groups = *v && *v != '0';
  if (const char *v = getenv("PLL_AMD_NO_FUSE"))
    if (*v && *v != '0') groups = false;
  c->fuse_mfma = groups && g.rate_cats <= 16 && g.states >= 17 && g.states <= 32;
  c->lean = !c->dna_fast && !c->use_mfma && g.states >= 17 && g.states <= 20 && g.rate_cats <= 4;
  if (const char *v = getenv("PLL_AMD_NO_LEAN")) // A/B switch: the scalar-fed FMA kernels for these shapes
    if (*v && *v != '0') c->lean = false;
  c->pm_stride = (size_t)g.rate_cats * g.states * gg.SPT;
  c->span = g.rate_cats * g.states_padded;
}

// A node that holds fewer entries than sites is class-compressed (site repeats): its CLV stays
// entry-contiguous on the device, in the host's own layout [entry][rate][states_padded] - gathers then read
// whole entries (kernels_dna.h, kernels_generic.h). The matrix-pipe kernels (33..64 states) keep the tiled layout.
static inline bool aos_entries(const pllgpu_ctx *c, unsigned entries)
{
  return (c->dna_fast || (c->generic_aos && !c->use_mfma)) && entries != c->geo.sites_alloc;
}

// doubles of device storage for `entries` entries of one CLV
static inline size_t clv_elems(const pllgpu_ctx *c, unsigned entries)
{
  if (aos_entries(c, entries)) return (size_t)entries * c->span;
  if (c->tiled) return (size_t)((entries + 63u) / 64u) * c->gg.tile_sz;
  return (size_t)entries * c->span;
}

extern "C" pllgpu_ctx_t *pllgpu_create(const pllgpu_geometry_t *geo, int device)
{
  int n = pllgpu_device_count();
  if (n <= 0)
  {
    fail(PLLGPU_ENODEVICE, "no HIP device visible (hipGetDeviceCount = %d)", n);
    return nullptr;
  }
  if (device < 0)
  {
    // PLL_AMD_DEVICE=<n>: that device; PLL_AMD_DEVICE=auto: new partitions go round-robin over the visible
    // devices (thread-per-partition callers such as RAxML-NG use every GPU of the node without a code change);
    // unset: the calling thread's current HIP device
    const char *env = getenv("PLL_AMD_DEVICE");
    static std::atomic<unsigned> next_device{0};
    if (env && strcmp(env, "auto") == 0)
      device = (int)(next_device.fetch_add(1, std::memory_order_relaxed) % (unsigned)n);
    else if (env)
      device = atoi(env);
    else if (hipGetDevice(&device) != hipSuccess)
      device = 0;
  }
  if (device >= n)
  {
    fail(PLLGPU_EINVAL, "device %d out of range (%d visible)", device, n);
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
  {
    fail(PLLGPU_ENODEVICE, "hipGetDeviceProperties(%d) failed", device);
    return nullptr;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
  {
    fail(PLLGPU_ENODEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
         prop.gcnArchName);
    return nullptr;
  }
  if (geo->states < 2 || geo->states > 64 || geo->rate_cats < 1 || geo->rate_cats > (unsigned)kMaxRates)
  {
    fail(PLLGPU_EUNSUPPORTED, "unsupported shape: states=%u (2..64) rate_cats=%u (1..%d)", geo->states,
         geo->rate_cats, kMaxRates);
    return nullptr;
  }
  pllgpu_ctx *c = new pllgpu_ctx();
  c->geo = *geo;
  c->device = device;
  derive_geometry(c);
  DeviceScope device_scope_(c); // the caller's current device comes back when this function returns
  bool ok = device_scope_.rc == 0 && hipSetDevice(device) == hipSuccess &&
            hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
            hipHostMalloc((void **)&c->result_host, kResultBytes, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void **)&c->result_dev, c->result_host, 0) == hipSuccess &&
            hipHostMalloc((void **)&c->rep_host, (kRepHostCap + 2) * sizeof(unsigned), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void **)&c->rep_host_dev, c->rep_host, 0) == hipSuccess;
  c->rep_host_cap = kRepHostCap;
  if (!ok)
  {
    fail(PLLGPU_ERUNTIME, "stream/event creation failed: %s", hipGetErrorString(hipGetLastError()));
    pllgpu_destroy(c);
    return nullptr;
  }
  c->own_stream = true;
  memset(c->result_host, 0, 8 * sizeof(double));
  memset(c->rep_host, 0, (kRepHostCap + 2) * sizeof(unsigned));
  c->clv.resize(geo->nodes);
  c->clv_aos.assign(geo->nodes, 0);
  c->scaler.resize(geo->scale_buffers);
  c->tipchars.resize(geo->tips);
  c->site_id.resize(geo->nodes);
  c->id_site.resize(geo->nodes);
  c->site_id8.resize(geo->nodes);
  c->rep_keep.resize(geo->nodes);
  c->map_forms.assign(geo->nodes, 0);
  c->map_widened.assign(geo->nodes, 0);
  c->lent.resize(geo->nodes);
  c->rent.resize(geo->nodes);
  c->rep_left.assign(geo->nodes, -1);
  c->rep_right.assign(geo->nodes, -1);
  c->ids.assign(geo->nodes, 0);
  if (c->pmat.ensure(c->pm_stride * (geo->prob_matrices + 2)) || c->freqs.ensure((size_t)geo->rate_matrices * geo->states_padded) ||
      c->rate_weights.ensure(geo->rate_cats) || c->prop_invar.ensure(geo->rate_matrices) ||
      c->pattern_weights.ensure(geo->sites_alloc) || c->persite.ensure(geo->sites_alloc) ||
      c->block_sums.ensure(4096) || c->counter.ensure(4) || c->rep_sync.ensure((size_t)kRepOps + 1) || c->rep_changed.ensure(1))
  {
    pllgpu_destroy(c);
    return nullptr;
  }
  (void)hipMemsetAsync(c->pmat.p, 0, c->pmat.cap * sizeof(double), c->stream);
  c->pm_version.assign(geo->prob_matrices + 2, 1ull);
  (void)hipMemsetAsync(c->prop_invar.p, 0, c->prop_invar.cap * sizeof(double), c->stream);
  (void)hipMemsetAsync(c->counter.p, 0, c->counter.cap * sizeof(unsigned), c->stream);
  (void)hipMemsetAsync(c->rep_sync.p, 0, c->rep_sync.cap * sizeof(unsigned), c->stream);
  (void)hipMemsetAsync(c->rep_changed.p, 0, sizeof(unsigned), c->stream);
  if (const char *v = getenv("PLL_AMD_REP_WGS")) c->rep_wgs = (unsigned)std::max(0, atoi(v));
  if (const char *v = getenv("PLL_AMD_REP_BITS")) c->rep_bits = !(*v == '0');
  if (const char *v = getenv("PLL_AMD_SUB_PACK_ALWAYS")) c->sub_pack_always = *v && *v != '0';
  if (const char *v = getenv("PLL_AMD_REP_FUSE")) c->rep_fuse = !(*v == '0');
  if (const char *v = getenv("PLL_AMD_PINNED_STAGING")) c->ring_failed = *v == '0'; // 0: every transfer from / to pageable memory as the runtime does it (A/B)
  if (const char *v = getenv("PLL_AMD_REP_HINTS")) c->rep_hints = !(*v == '0');
  if (const char *v = getenv("PLL_AMD_REP_RANGES")) c->rep_max_ranges = (unsigned)std::max(1, atoi(v));
  return c;
}

static void drop_chain_plan(pllgpu_ctx *c);

extern "C" void pllgpu_destroy(pllgpu_ctx_t *c)
{
  if (!c) return;
  DeviceScope device_scope_(c);
  c->deferred.clear(); // results nobody will ask for
  c->chain_held = false;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drop_chain_plan(c);
  for (LevelPlan *lp : c->level_plans) delete lp;
  c->level_plans.clear();
  c->chain_dev.release();
  c->sub_dev.release();
  c->sub_packed.release();
  c->cherry_bits.release();
  for (auto &b : c->clv) b.release();
  for (auto &b : c->scaler) b.release();
  for (auto &b : c->tipchars) b.release();
  for (auto &b : c->site_id) b.release();
  for (auto &b : c->id_site) b.release();
  for (auto &b : c->site_id8) b.release();
  for (auto &b : c->rep_keep) b.release();
  for (auto &b : c->lent) b.release();
  for (auto &b : c->rent) b.release();
  c->tipmap.release();
  c->scratch.release();
  c->pmat.release();
  c->edge_partials.release();
  c->edge_tickets.release();
  c->freqs.release();
  c->rate_weights.release();
  c->prop_invar.release();
  c->persite.release();
  c->block_sums.release();
  c->reduce.release();
  c->counter.release();
  c->mfma_flags.release();
  c->eigenvals.release();
  c->evecs.release();
  c->ievecs.release();
  c->brlen.release();
  c->mindex.release();
  c->rep_table.release();
  c->rep_ops.release();
  c->rep_blocksum.release();
  c->rep_counts.release();
  c->rep_sync.release();
  c->rep_changed.release();
  c->rep_final.release();
  c->rates.release();
  c->diag.release();
  for (auto &b : c->sumtable) b.release();
  c->pattern_weights.release();
  c->invariant.release();
  c->pool.trim(0); // everything above went to the context's idle blocks: back to the runtime now
  if (c->result_host) (void)hipHostFree(c->result_host);
  if (c->rep_host) (void)hipHostFree(c->rep_host);
  if (c->ring_host) (void)hipHostFree(c->ring_host);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ---- data movement ---------------------------------------------------------------------------
// H2D copies from pageable caller memory: hipMemcpyAsync stages pageable sources before it
// returns, so the caller may reuse its buffer; ordering with kernels is by the stream.
// every entry point first launches what pllgpu_update_partials is holding back (tail fusion,
// kernels_dna.h) - except the edge evaluation, which may consume it
static int flush_deferred(pllgpu_ctx *c);
#define CHECK_CTX_KEEP(c)                    \
  if (!(c)) return fail(PLLGPU_EINVAL, "null context"); \
  DeviceScope device_scope_(c);              \
  if (device_scope_.rc) return device_scope_.rc
#define CHECK_CTX(c)                                   \
  CHECK_CTX_KEEP(c);                                   \
  if (!(c)->deferred.empty() || (c)->chain_held)       \
    if (int rc_ = flush_deferred(c)) return rc_

// Small transfers. hipMemcpyAsync from or to pageable memory is a blocking trip through the runtime's own staging (10-25 us
// whatever the size; the flat pll_core_* seam made six of them per call): a transfer of up to kRingMax bytes goes through
// the context's block of pinned host memory instead - up: the caller's bytes are copied there and the device reads them
// from there (a copy enqueued from pinned memory, or the layout kernel straight out of host memory); down: the device
// writes there, one wait, the bytes are copied out. A piece of the block belongs to its transfer until the stream has
// been waited for; the block is handed out front to back and starts over after a wait (stream_wait).
constexpr size_t kRingCap = (size_t)8 << 20, kRingMax = (size_t)2 << 20;

// the stream has just been waited for: everything that read or wrote the block has completed
static void ring_drained(pllgpu_ctx *c)
{
  for (const auto &d : c->pending_down) memcpy(d.host, d.ring, d.bytes);
  c->pending_down.clear();
  c->ring_off = 0;
}

static hipError_t stream_wait(pllgpu_ctx *c)
{
  const hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) ring_drained(c);
  else
  {
    // the downloads that were waiting for this did not happen: their host pointers (a caller's arrays, perhaps out of
    // scope by the next wait) are forgotten, and the block is not handed out again - what the device still does with
    // it is unknown
    c->pending_down.clear();
    c->ring_failed = true;
  }
  return e;
}

// `bytes` of the block: the host address, and the address the device knows the same memory by. nullptr: not this way
// (too large, or no pinned memory to be had) - the caller takes the pageable path.
static unsigned char *stage_take(pllgpu_ctx *c, size_t bytes, unsigned char **dev)
{
  if (bytes > kRingMax || c->ring_failed) return nullptr;
  if (!c->ring_host)
  {
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, kRingCap, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess)
    {
      if (h) (void)hipHostFree(h);
      (void)hipGetLastError();
      c->ring_failed = true;
      return nullptr;
    }
    c->ring_host = (unsigned char *)h;
    c->ring_dev = (unsigned char *)d;
    c->ring_cap = kRingCap;
    c->ring_off = 0;
  }
  const size_t need = (bytes + 255u) & ~(size_t)255u;
  if (c->ring_off + need > c->ring_cap && stream_wait(c) != hipSuccess) return nullptr;
  unsigned char *h = c->ring_host + c->ring_off;
  *dev = c->ring_dev + c->ring_off;
  c->ring_off += need;
  return h;
}

// host -> device, `bytes` from pageable caller memory
static hipError_t copy_up(pllgpu_ctx *c, void *dst, const void *host, size_t bytes)
{
  unsigned char *dev = nullptr;
  if (unsigned char *h = stage_take(c, bytes, &dev))
  {
    memcpy(h, host, bytes);
    return hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, c->stream);
  }
  return hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, c->stream);
}

// device -> host and wait
static hipError_t copy_down(pllgpu_ctx *c, void *host, const void *src, size_t bytes)
{
  unsigned char *dev = nullptr;
  if (unsigned char *h = stage_take(c, bytes, &dev))
  {
    hipError_t e = hipMemcpyAsync(h, src, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return e;
    c->pending_down.push_back({host, h, bytes});
    return c->defer_down ? hipSuccess : stream_wait(c);
  }
  const hipError_t e = hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, c->stream);
  return e == hipSuccess ? stream_wait(c) : e;
}

extern "C" int pllgpu_download_defer(pllgpu_ctx_t *c, int on)
{
  CHECK_CTX_KEEP(c);
  c->defer_down = on != 0;
  if (!on && !c->pending_down.empty()) HIP_TRY(stream_wait(c));
  return 0;
}

extern "C" int pllgpu_clv_reserve(pllgpu_ctx_t *c, unsigned node, unsigned entries)
{
  CHECK_CTX(c);
  if (node >= c->geo.nodes) return fail(PLLGPU_EINVAL, "clv index %u out of range", node);
  return c->clv[node].ensure(clv_elems(c, entries));
}

extern "C" int pllgpu_clv_upload(pllgpu_ctx_t *c, unsigned node, const double *host, unsigned entries)
{
  CHECK_CTX(c); // (the context's device for the copies and the layout kernel below, not only for the reservation)
  if (int rc = pllgpu_clv_reserve(c, node, entries)) return rc;
  const size_t bytes = (size_t)entries * c->span * sizeof(double);
  c->clv_aos[node] = aos_entries(c, entries) ? 1 : 0;
  if (!c->tiled || c->clv_aos[node])
  {
    HIP_TRY(copy_up(c, c->clv[node].p, host, bytes));
    return 0;
  }
  unsigned char *dev = nullptr;
  if (unsigned char *h = stage_take(c, bytes, &dev))
  {
    // the layout kernel reads the caller's entries out of host memory
    memcpy(h, host, bytes);
    const unsigned blocks = (unsigned)std::min<size_t>(1024, ((size_t)(entries + 63u) * c->span + 255u) / 256u);
    hipLaunchKernelGGL(k_host_aos_to_tiled, dim3(blocks), dim3(256), 0, c->stream, reinterpret_cast<const double *>(dev), c->clv[node].p, entries,
                       c->gg.S, c->gg.SP, c->gg.R);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (int rc = c->scratch.ensure((size_t)entries * c->span)) return rc;
  HIP_TRY(hipMemcpyAsync(c->scratch.p, host, bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_aos_to_tiled, dim3(1024), dim3(256), 0, c->stream, c->scratch.p, c->clv[node].p, entries,
                     c->gg.S, c->gg.SP, c->gg.R);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int pllgpu_clv_download(pllgpu_ctx_t *c, unsigned node, double *host, unsigned entries)
{
  CHECK_CTX(c);
  if (node >= c->geo.nodes || clv_elems(c, entries) > c->clv[node].cap)
    return fail(PLLGPU_EINVAL, "clv %u: download of %u entries exceeds the device buffer", node, entries);
  const size_t bytes = (size_t)entries * c->span * sizeof(double);
  const double *src = c->clv[node].p;
  if (c->tiled && !c->clv_aos[node])
  {
    unsigned char *dev = nullptr;
    if (unsigned char *h = stage_take(c, bytes, &dev))
    {
      // the layout kernel writes the entries into host memory
      const unsigned blocks = (unsigned)std::min<size_t>(1024, ((size_t)entries * c->span + 255u) / 256u);
      hipLaunchKernelGGL(k_host_tiled_to_aos, dim3(blocks), dim3(256), 0, c->stream, c->clv[node].p, reinterpret_cast<double *>(dev), entries,
                         c->gg.S, c->gg.SP, c->gg.R);
      HIP_TRY(hipGetLastError());
      c->pending_down.push_back({host, h, bytes});
      if (!c->defer_down) HIP_TRY(stream_wait(c));
      return 0;
    }
    if (int rc = c->scratch.ensure((size_t)entries * c->span)) return rc;
    hipLaunchKernelGGL(k_tiled_to_aos, dim3(1024), dim3(256), 0, c->stream, c->clv[node].p, c->scratch.p, entries,
                       c->gg.S, c->gg.SP, c->gg.R);
    HIP_TRY(hipGetLastError());
    src = c->scratch.p;
  }
  HIP_TRY(copy_down(c, host, src, bytes));
  return 0;
}

static inline size_t scaler_elems(const pllgpu_ctx *c, unsigned entries)
{
  return (size_t)entries * (c->geo.per_rate_scalers ? c->geo.rate_cats : 1u);
}

extern "C" int pllgpu_scaler_reserve(pllgpu_ctx_t *c, unsigned index, unsigned entries)
{
  CHECK_CTX(c);
  if (index >= c->geo.scale_buffers) return fail(PLLGPU_EINVAL, "scale buffer %u out of range", index);
  return c->scaler[index].ensure(scaler_elems(c, entries));
}

extern "C" int pllgpu_scaler_upload(pllgpu_ctx_t *c, unsigned index, const unsigned *host, unsigned entries)
{
  CHECK_CTX(c);
  if (int rc = pllgpu_scaler_reserve(c, index, entries)) return rc;
  HIP_TRY(copy_up(c, c->scaler[index].p, host, scaler_elems(c, entries) * sizeof(unsigned)));
  return 0;
}

extern "C" int pllgpu_scaler_download(pllgpu_ctx_t *c, unsigned index, unsigned *host, unsigned entries)
{
  CHECK_CTX(c);
  if (index >= c->geo.scale_buffers || scaler_elems(c, entries) > c->scaler[index].cap)
    return fail(PLLGPU_EINVAL, "scale buffer %u: download exceeds the device buffer", index);
  HIP_TRY(copy_down(c, host, c->scaler[index].p, scaler_elems(c, entries) * sizeof(unsigned)));
  return 0;
}

extern "C" int pllgpu_tipchars_upload(pllgpu_ctx_t *c, unsigned tip, const unsigned char *host, unsigned count)
{
  CHECK_CTX(c);
  if (tip >= c->geo.tips) return fail(PLLGPU_EINVAL, "tip %u out of range", tip);
  if (int rc = c->tipchars[tip].ensure(((size_t)count + 65) & ~(size_t)63)) return rc; // kernels fetch two codes at a time
  HIP_TRY(hipMemcpyAsync(c->tipchars[tip].p, host, count, hipMemcpyHostToDevice, c->stream));
  ++c->tips_epoch;
  return 0;
}

extern "C" int pllgpu_tipmap_upload(pllgpu_ctx_t *c, const unsigned long long *host, unsigned count)
{
  CHECK_CTX(c);
  if (!host)
  {
    c->tipmap_set = false;
    return 0;
  }
  if (int rc = c->tipmap.ensure(256)) return rc;
  HIP_TRY(hipMemcpyAsync(c->tipmap.p, host, std::min(count, 256u) * sizeof(unsigned long long),
                         hipMemcpyHostToDevice, c->stream));
  c->tipmap_set = true;
  c->tip_ncodes = 0;
  for (unsigned i = 0; i < std::min(count, 256u); ++i)
    if (host[i]) c->tip_ncodes = i + 1;
  ++c->maps_epoch; // cached launches carry the code count
  return 0;
}

// matrices [first, first+count) of the device block; slots prob_matrices and prob_matrices+1 hold the
// two sumtable contraction matrices (kernels_deriv.h)
static int upload_matrices(pllgpu_ctx *c, unsigned first, unsigned count, const double *host, unsigned limit)
{
  const pllgpu_geometry_t &g = c->geo;
  if (first + count > limit) return fail(PLLGPU_EINVAL, "matrix range [%u,%u) out of range", first, first + count);
  const unsigned S = g.states, SP = g.states_padded, R = g.rate_cats, SPT = c->gg.SPT;
  const size_t host_stride = (size_t)R * S * SP;
  // the device's layout is formed in pinned memory where the block fits (no wait), else in the staging vector
  const size_t doubles = c->pm_stride * count;
  unsigned char *dev = nullptr;
  double *stage = reinterpret_cast<double *>(stage_take(c, doubles * sizeof(double), &dev));
  if (stage)
    memset(stage, 0, doubles * sizeof(double));
  else
  {
    HIP_TRY(stream_wait(c)); // the previous async copy may still be reading the staging vector
    c->stage.assign(doubles, 0.0);
    stage = c->stage.data();
  }
  for (unsigned m = 0; m < count; ++m)
    for (unsigned k = 0; k < R; ++k)
      for (unsigned i = 0; i < S; ++i)
      {
        const double *row = host + m * host_stride + ((size_t)k * S + i) * SP;
        double *dst = stage + m * c->pm_stride + (size_t)k * S * SPT + i;
        for (unsigned j = 0; j < S; ++j) dst[(size_t)j * SPT] = row[j];
      }
  HIP_TRY(hipMemcpyAsync(c->pmat.p + (size_t)first * c->pm_stride, stage, doubles * sizeof(double), hipMemcpyHostToDevice, c->stream));
  for (unsigned m = first; m < first + count && m < c->pm_version.size(); ++m) ++c->pm_version[m];
  return 0;
}

extern "C" int pllgpu_pmatrix_upload(pllgpu_ctx_t *c, unsigned first, unsigned count, const double *host)
{
  CHECK_CTX(c);
  return upload_matrices(c, first, count, host, c->geo.prob_matrices);
}

extern "C" int pllgpu_aux_matrix_upload(pllgpu_ctx_t *c, unsigned slot, const double *host)
{
  CHECK_CTX(c);
  if (slot > 1) return fail(PLLGPU_EINVAL, "aux matrix slot %u out of range", slot);
  return upload_matrices(c, c->geo.prob_matrices + slot, 1, host, c->geo.prob_matrices + 2);
}

extern "C" int pllgpu_frequencies_upload(pllgpu_ctx_t *c, unsigned index, const double *host)
{
  CHECK_CTX(c);
  if (index >= c->geo.rate_matrices) return fail(PLLGPU_EINVAL, "frequency set %u out of range", index);
  HIP_TRY(hipMemcpyAsync(c->freqs.p + (size_t)index * c->geo.states_padded, host,
                         c->geo.states_padded * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}

extern "C" int pllgpu_rate_weights_upload(pllgpu_ctx_t *c, const double *host)
{
  CHECK_CTX(c);
  HIP_TRY(hipMemcpyAsync(c->rate_weights.p, host, c->geo.rate_cats * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}

extern "C" int pllgpu_prop_invar_upload(pllgpu_ctx_t *c, const double *host)
{
  CHECK_CTX(c);
  HIP_TRY(hipMemcpyAsync(c->prop_invar.p, host, c->geo.rate_matrices * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}

extern "C" int pllgpu_pattern_weights_upload(pllgpu_ctx_t *c, const unsigned *host, unsigned count)
{
  CHECK_CTX(c);
  if (count > c->geo.sites_alloc) return fail(PLLGPU_EINVAL, "pattern weight count %u > %u", count, c->geo.sites_alloc);
  HIP_TRY(hipMemcpyAsync(c->pattern_weights.p, host, count * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
  return 0;
}

extern "C" int pllgpu_invariant_upload(pllgpu_ctx_t *c, const int *host, unsigned count)
{
  CHECK_CTX(c);
  if (!host)
  {
    c->invariant_set = false;
    return 0;
  }
  if (int rc = c->invariant.ensure(count)) return rc;
  HIP_TRY(hipMemcpyAsync(c->invariant.p, host, count * sizeof(int), hipMemcpyHostToDevice, c->stream));
  c->invariant_set = true;
  return 0;
}

// entries of a site -> class map buffer: whole groups of sixteen (kernels_repeats.h reads and writes them so)
static inline size_t map_elems(const pllgpu_ctx *c) { return ((size_t)c->geo.sites_alloc + 15u) / 16u * 16u; }

extern "C" int pllgpu_repeats_upload(pllgpu_ctx_t *c, unsigned node, const unsigned *site_id,
                                     const unsigned *id_site, unsigned ids)
{
  CHECK_CTX(c);
  if (node >= c->geo.nodes) return fail(PLLGPU_EINVAL, "node %u out of range", node);
  c->ids[node] = ids;
  ++c->maps_epoch;
  ++c->maps_version;
  ++c->maps_foreign;
  c->map_widened[node] = 0;
  c->rep_left[node] = c->rep_right[node] = -1; // host-built maps: no entry-indexed child maps
  c->map_forms[node] = 0;
  if (!ids) return 0;
  if (int rc = c->id_site[node].ensure(ids)) return rc;
  if (id_site) HIP_TRY(copy_up(c, c->id_site[node].p, id_site, ids * sizeof(unsigned)));
  if (ids <= kRepNarrow)
  {
    // at most 256 classes - every tip: the map goes up as BYTES (a quarter of the bus: 128 tips x 1M sites were 512 MB), the
    // form the class kernels read; the 32-bit form the site-indexed gathers and the API want is made on the device on demand
    if (int rc = c->site_id8[node].ensure(map_elems(c))) return rc;
    c->stage8.resize(c->geo.sites_alloc);
    for (unsigned s = 0; s < c->geo.sites_alloc; ++s) c->stage8[s] = (unsigned char)site_id[s];
    HIP_TRY(hipMemcpyAsync(c->site_id8[node].p, c->stage8.data(), c->geo.sites_alloc, hipMemcpyHostToDevice, c->stream)); // (pageable: staged before it returns)
    c->map_forms[node] = kMap8;
    return 0;
  }
  if (int rc = c->site_id[node].ensure(map_elems(c))) return rc;
  HIP_TRY(hipMemcpyAsync(c->site_id[node].p, site_id, c->geo.sites_alloc * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
  c->map_forms[node] = kMap32;
  return 0;
}

// The 32-bit form of a node's site -> class map, produced from the byte form if that is what the class kernels left
// (kernels_repeats.h); nullptr if the node has no maps on the device. Callers: the gathering launches that index by site,
// the edge / root / derivative evaluations, the copy to the host.
static const unsigned *wide_map(pllgpu_ctx *c, unsigned node)
{
  if (c->map_forms[node] & kMap32) return c->site_id[node].p;
  if (!(c->map_forms[node] & kMap8)) return c->site_id[node].p; // (uploaded by an older path or never written: as before)
  if (c->site_id[node].ensure(map_elems(c))) return nullptr;
  const unsigned n = (unsigned)map_elems(c);
  hipLaunchKernelGGL(k_rep_widen, dim3((n / 16u + 255u) / 256u), dim3(256), 0, c->stream, c->site_id8[node].p, c->site_id[node].p, n);
  c->map_forms[node] |= kMap32;
  c->map_widened[node] = 1; // from now on the class kernels' bytes are followed by this form again (pllgpu_repeats_classes)
  return c->site_id[node].p;
}

static int narrow_map(pllgpu_ctx *c, unsigned node)
{
  if (c->map_forms[node] & kMap8) return 0;
  if (!(c->map_forms[node] & kMap32) || !c->site_id[node].p) return fail(PLLGPU_EINVAL, "node %u has no class maps on the device", node);
  if (int rc = c->site_id8[node].ensure(map_elems(c))) return rc;
  const unsigned n = (unsigned)map_elems(c);
  hipLaunchKernelGGL(k_rep_narrow, dim3((n / 16u + 255u) / 256u), dim3(256), 0, c->stream, c->site_id[node].p, c->site_id8[node].p, n);
  c->map_forms[node] |= kMap8;
  return 0;
}

// ---- launches --------------------------------------------------------------------------------
static int resolve_op(pllgpu_ctx *c, const pllgpu_op_t &o, DevOp &d)
{
  const pllgpu_geometry_t &g = c->geo;
  if (o.parent_clv >= g.nodes || o.left_clv >= g.nodes || o.right_clv >= g.nodes)
    return fail(PLLGPU_EINVAL, "operation references a CLV out of range");
  if (o.left_matrix >= g.prob_matrices || o.right_matrix >= g.prob_matrices)
    return fail(PLLGPU_EINVAL, "operation references a p-matrix out of range");
  memset(&d, 0, sizeof d);
  if (int rc = c->clv[o.parent_clv].ensure(clv_elems(c, o.parent_entries))) return rc;
  d.parent = c->clv[o.parent_clv].p;
  d.entries = o.parent_entries;
  // layout of the three CLVs (4x4 only): what the children were written as, what the parent becomes
  c->clv_aos[o.parent_clv] = aos_entries(c, o.parent_entries) ? 1 : 0;
  if (c->clv_aos[o.parent_clv]) c->any_aos = true;
  d.layout = c->clv_aos[o.parent_clv] ? kAosParent : 0u;
  if (!(o.flags & PLLGPU_OP_LEFT_TIP) && c->clv_aos[o.left_clv]) d.layout |= kAosLeft;
  if (!(o.flags & PLLGPU_OP_RIGHT_TIP) && c->clv_aos[o.right_clv]) d.layout |= kAosRight;
  if (d.layout && !(o.flags & PLLGPU_OP_GATHER)) return fail(PLLGPU_EINVAL, "a class-compressed CLV met an operation without the gather flag");
  // a compressed child with about as many entries as the parent is read once per entry: streaming loads;
  // a much smaller one is a table the parent's tiles keep coming back to: cacheable loads
  if ((d.layout & kAosLeft) && (size_t)c->ids[o.left_clv] * 2u > o.parent_entries) d.layout |= kStreamLeft;
  if ((d.layout & kAosRight) && (size_t)c->ids[o.right_clv] * 2u > o.parent_entries) d.layout |= kStreamRight;
  if (o.flags & PLLGPU_OP_LEFT_TIP)
  {
    if (o.left_clv >= g.tips || !c->tipchars[o.left_clv].p) return fail(PLLGPU_EINVAL, "tip %u has no codes on the device", o.left_clv);
    d.ltip = c->tipchars[o.left_clv].p;
  }
  else
  {
    if (!c->clv[o.left_clv].p) return fail(PLLGPU_EINVAL, "CLV %u was never computed or uploaded", o.left_clv);
    d.left = c->clv[o.left_clv].p;
  }
  if (o.flags & PLLGPU_OP_RIGHT_TIP)
  {
    if (o.right_clv >= g.tips || !c->tipchars[o.right_clv].p) return fail(PLLGPU_EINVAL, "tip %u has no codes on the device", o.right_clv);
    d.rtip = c->tipchars[o.right_clv].p;
  }
  else
  {
    if (!c->clv[o.right_clv].p) return fail(PLLGPU_EINVAL, "CLV %u was never computed or uploaded", o.right_clv);
    d.right = c->clv[o.right_clv].p;
  }
  auto scal = [&](int idx, const unsigned *&out, unsigned need) -> int {
    out = nullptr;
    if (idx < 0) return 0;
    if ((unsigned)idx >= g.scale_buffers) return fail(PLLGPU_EINVAL, "scale buffer %d out of range", idx);
    if (need)
    {
      if (int rc = c->scaler[idx].ensure(scaler_elems(c, need))) return rc;
    }
    else if (!c->scaler[idx].p)
      return fail(PLLGPU_EINVAL, "scale buffer %d read before it was written", idx);
    out = c->scaler[idx].p;
    return 0;
  };
  const unsigned *ps = nullptr;
  if (int rc = scal(o.parent_scaler, ps, o.parent_entries)) return rc;
  d.pscaler = const_cast<unsigned *>(ps);
  if (int rc = scal(o.left_scaler, d.lscaler, 0)) return rc;
  if (int rc = scal(o.right_scaler, d.rscaler, 0)) return rc;
  d.lmat = c->pmat.p + (size_t)o.left_matrix * c->pm_stride;
  d.rmat = c->pmat.p + (size_t)o.right_matrix * c->pm_stride;
  if (o.flags & PLLGPU_OP_GATHER)
  {
    const unsigned p = o.parent_clv;
    if (c->ids[p] && c->rep_left[p] == (int)o.left_clv && c->rep_right[p] == (int)o.right_clv)
    {
      d.lsid = c->lent[p].p; // entry-indexed: two coalesced loads per lane
      d.rsid = c->rent[p].p;
      d.layout |= kDirectMaps;
    }
    else if (c->ids[p] && c->rep_left[p] == (int)o.right_clv && c->rep_right[p] == (int)o.left_clv)
    {
      d.lsid = c->rent[p].p; // the host put the tip on the left
      d.rsid = c->lent[p].p;
      d.layout |= kDirectMaps;
    }
    else
    {
      d.id_site = c->ids[p] ? c->id_site[p].p : nullptr;
      d.lsid = c->ids[o.left_clv] ? wide_map(c, o.left_clv) : nullptr;
      d.rsid = c->ids[o.right_clv] ? wide_map(c, o.right_clv) : nullptr;
      if ((c->ids[o.left_clv] && !d.lsid) || (c->ids[o.right_clv] && !d.rsid)) return fail(PLLGPU_EINVAL, "operation gathers through class maps that are not on the device");
    }
  }
  return 0;
}

template <int ICH>
static void launch_generic(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned maxent, unsigned kind, bool gather)
{
  // one workgroup per 64-entry tile; min(R,4) waves share the tile's rate categories
  const unsigned tiles = (maxent + 63) / 64;
  const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
  // tip children: room in LDS for the staged matrices of every wave (kernels_generic.h: tip_stage)
  const unsigned nw = std::min(c->gg.R, 4u);
  const bool stage = kind != 0 && c->gg.S * c->gg.SPT <= 1024u && !c->no_tip_columns;
  const unsigned tip_lds = stage ? 1u : 0u;
  size_t lds = stage ? (size_t)2 * nw * (c->gg.S + 1u) * (c->gg.SPT | 1u) * sizeof(double) : 0;
  // entry-contiguous parents leave through LDS (kernels_generic.h): 16-byte rows need states_padded % 4 == 0
  // and entry-contiguous children arrive through it, SP / 2 lanes per (entry, rate)
  bool any_aos = false;
  for (unsigned i = 0; i < nops; ++i) any_aos = any_aos || (pack.ops[i].layout & (kAosParent | kAosLeft | kAosRight));
  const unsigned par_lds = (gather && any_aos && c->gg.SP % 4u == 0 && (ICH != 20 || c->gg.SP == 20u) && !c->no_par_lds) ? (c->no_coop_fetch ? 2u : 1u) : 0u;
  if (par_lds) lds += (size_t)nw * 64u * (c->gg.SP + 2u) * sizeof(double);

  // staged tip matrices are shared by the tiles of a workgroup: several tiles each, as long as
  // ~2048 workgroups remain
  unsigned tpb = 1;
  // (staging costs a workgroup about as much as four tiles of work: C3's 32-op launch 259 us with up to 8
  // tiles per workgroup and >= 4096 workgroups, 247 us with up to 16 and >= 2048, 268 us with 32 / 1024)
  if (stage) tpb = std::max(1u, std::min(16u, (unsigned)(((size_t)tiles * nops) / 2048u)));
  dim3 grid((tiles + tpb - 1) / tpb, nops), block(64u * nw);
  if (gather && ICH == 20)
  {
    // the protein shape: when every op of the launch has the same CLV layouts (the rule: a level of a tree), the
    // instantiation that knows them at compile time
    const unsigned lay = pack.ops[0].layout & 7u;
    bool same = true;
    for (unsigned i = 1; i < nops; ++i) same = same && (pack.ops[i].layout & 7u) == lay;
    if (same)
    {
#define GEN_LAY(LT, RT, L) \
  case L: if (par_lds) raise_lds_limit((const void *)k_partials_tiled<ICH == 20 ? 20 : ICH, LT, RT, true, ICH == 20 ? L : -1>, c->device, lds); \
    hipLaunchKernelGGL((k_partials_tiled<ICH == 20 ? 20 : ICH, LT, RT, true, ICH == 20 ? L : -1>), grid, block, lds, c->stream, pack, c->gg, tm, tip_lds, tpb, par_lds); return;
#define GEN_LAYS(LT, RT) \
  switch (lay)             \
  {                        \
    GEN_LAY(LT, RT, 0) GEN_LAY(LT, RT, 1) GEN_LAY(LT, RT, 2) GEN_LAY(LT, RT, 3) GEN_LAY(LT, RT, 4) GEN_LAY(LT, RT, 5) GEN_LAY(LT, RT, 6) GEN_LAY(LT, RT, 7) \
  }
      if (kind == 0) { GEN_LAYS(false, false) }
      else if (kind == 1) { GEN_LAYS(true, false) }
      else { GEN_LAYS(true, true) }
#undef GEN_LAYS
#undef GEN_LAY
    }
  }
#define GEN_LAUNCH(LT, RT, GA) \
  do { if (par_lds) raise_lds_limit((const void *)k_partials_tiled<ICH, LT, RT, GA>, c->device, lds); \
  hipLaunchKernelGGL((k_partials_tiled<ICH, LT, RT, GA>), grid, block, lds, c->stream, pack, c->gg, tm, tip_lds, tpb, par_lds); } while (0)
  if (kind == 0)
  {
    if (gather) GEN_LAUNCH(false, false, true); else GEN_LAUNCH(false, false, false);
  }
  else if (kind == 1)
  {
    if (gather) GEN_LAUNCH(true, false, true); else GEN_LAUNCH(true, false, false);
  }
  else
  {
    if (gather) GEN_LAUNCH(true, true, true); else GEN_LAUNCH(true, true, false);
  }
#undef GEN_LAUNCH
}

static void launch_dna(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned maxent, unsigned kind, bool gather)
{
  // one wave per 64-site tile (kDnaTilesPerWave, kernels_dna.h)
  const unsigned tiles = (maxent + 63) / 64;
  const unsigned tpw = kDnaTilesPerWave;
  const unsigned nx = (tiles + 4 * tpw - 1) / (4 * tpw);
  dim3 grid = xcd_grid(nx, nops), block(256);
  const int mode = c->gg.scale_mode;
  // gathering launches: a pattern-sorted alignment's neighbouring sites gather neighbouring entries, and with the XCD-aware
  // order (kernels_common.h) an XCD's L2 holds the entries of ITS run of sites, not a copy of everybody's
  const unsigned xcd = (c->xcd_order && gather) ? 1u : 0u;
#define DNA_LAUNCH(LT, RT, GA) hipLaunchKernelGGL((k_partials_dna<LT, RT, GA>), grid, block, 0, c->stream, pack, mode, tpw, nx, nops, xcd)
  if (kind == 0)
  {
    if (gather) DNA_LAUNCH(false, false, true); else DNA_LAUNCH(false, false, false);
  }
  else if (kind == 1)
  {
    if (gather) DNA_LAUNCH(true, false, true); else DNA_LAUNCH(true, false, false);
  }
  else
  {
    if (gather) DNA_LAUNCH(true, true, true); else DNA_LAUNCH(true, true, false);
  }
#undef DNA_LAUNCH
}

// site repeats: (gathering, gathering -> inner x inner) groups (kernels_dna.h: k_partials_dna_gg)
static void launch_gg(pllgpu_ctx *c, const GGPack &pack, unsigned ngroups, unsigned entries)
{
  const unsigned tiles = (entries + 63) / 64;
  const unsigned tpw = kDnaTilesPerWave;
  const unsigned nx = (tiles + 4 * tpw - 1) / (4 * tpw);
  dim3 grid = xcd_grid(nx, ngroups), block(256);
  const unsigned stream_parent = ((size_t)ngroups * entries * 128u > c->stream_parent_bytes) ? 1u : 0u;
  const unsigned xcd = c->xcd_order; // (C4's shard, same box: 0.1104-0.1111 -> 0.1089-0.1102 ms per step for the slowest shards)
  hipLaunchKernelGGL(k_partials_dna_gg, grid, block, 0, c->stream, pack, entries, c->gg.scale_mode, tpw, stream_parent, nx, ngroups, xcd);
}

// fp64 MFMA 4x4x4 kernels, matrices staged in LDS (kernels_mfma.h): NG = number of 4-state groups
// k_partials_mfma_wide counts its own vector-memory operations (kernels_mfma_wide.h): a register spilled to scratch
// would be one the count does not know about. The build is checked once per process; a compiler that spills sends the
// shape back to the first-generation kernel (and says so). This run-time look sees scratch only; what a compiler could
// also do - copy, move or park in accumulation registers a load's destination before its wait - is checked where it
// can be seen, on the emitted instructions at build time (tools/check_wide_isa.py, a prerequisite of libpll_amd.so in
// the Makefile; tests/test_wide_isa_check.py plants those failures into the real instruction stream).
template <int NGJ, int TAIL, int WAVES>
static bool wide_kernel_is_sound()
{
  static int st = -1;
  if (st < 0)
  {
    hipFuncAttributes at;
    st = (hipFuncGetAttributes(&at, (const void *)k_partials_mfma_wide<NGJ, TAIL, WAVES>) == hipSuccess && at.localSizeBytes == 0) ? 1 : 0;
    if (!st) fprintf(stderr, "libpll_amd: k_partials_mfma_wide was built with scratch memory; using k_partials_mfma instead\n");
  }
  return st == 1;
}

// One round of workgroups, one (8 waves) or two (4 waves) per CU; every (op, rate category) gets the same number of
// them and cuts its half tiles into that many runs.
template <int NGJ, int TAIL, int WAVES>
static bool launch_wide(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned halves, unsigned char *fb, unsigned fstride)
{
  if (!wide_kernel_is_sound<NGJ, TAIL, WAVES>()) return false;
  const unsigned R = c->gg.R, max_wgs = 2048u / WAVES;
  const unsigned per_pair = std::max(1u, max_wgs / (nops * R));
  const unsigned hw = std::max(1u, (halves + per_pair - 1) / per_pair);
  dim3 grid((halves + hw - 1) / hw, nops, R), block(64 * WAVES);
  const size_t lds = MfmaGeo<16>::lds_doubles * sizeof(double);
  raise_lds_limit((const void *)k_partials_mfma_wide<NGJ, TAIL, WAVES>, c->device, lds);
  hipLaunchKernelGGL((k_partials_mfma_wide<NGJ, TAIL, WAVES>), grid, block, lds, c->stream, pack, c->gg, hw, fb, fstride);
  return true;
}

static inline unsigned tt_stream_ld_host(unsigned S) { return (S + 1u) | 1u; } // (kernels_mfma.h: tt_stream_ld)

template <int NG>
static int launch_mfma_t(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned maxent, unsigned kind, bool gather)
{
  const unsigned R = c->gg.R;
  const unsigned items = (maxent + 31) / 32; // 32 sites per item
  if (NG == 16 && kind == 0 && !gather && c->mfma_wide)
  {
    // 33..64 states, inner x inner, tiled CLVs: the second-generation kernel (kernels_mfma_wide.h). Work is dealt in
    // half tiles: every SIMD gets two waves and the pair the same number of half tiles, give or take one
    bool scaling = false;
    for (unsigned i = 0; i < nops; ++i) scaling = scaling || pack.ops[i].pscaler != nullptr;
    scaling = scaling && c->gg.scale_mode != 0;
    const unsigned fstride = (maxent + 63u) & ~63u;
    if (c->mfma_flags.ensure(std::max<size_t>(64, scaling ? (size_t)kMaxOpsPerLaunch * R * fstride : 0))) return PLLGPU_ENOMEM;
    unsigned char *fb = c->mfma_flags.p;
    const bool exact61 = c->gg.S == 61 && !c->mfma_pad;
    // (8-wave workgroups, one per CU: against two of 4 waves the matrices are staged once per CU and a SIMD's two waves
    // split an odd share - C5's 2-op launch 66 -> 58 us, the step 0.549 -> 0.537 ms, profiles/README.md round 3)
    const bool done = exact61 ? launch_wide<15, 1, 8>(c, pack, nops, items, fb, fstride) : launch_wide<16, 0, 8>(c, pack, nops, items, fb, fstride);
    if (done)
    {
      if (scaling)
      {
        dim3 eg((maxent + 255) / 256, nops);
        hipLaunchKernelGGL((k_mfma_scale_epilogue<false>), eg, dim3(256), 0, c->stream, pack, c->gg, fb, fstride);
      }
      return 0;
    }
  }
  if (kind == 2 && !gather && NG > 8 && c->tt_stream && !c->no_tip_columns)
  {
    // plain tip x tip level of a large state space: a store stream, lane = site (kernels_mfma.h: k_partials_tt_stream)
    const unsigned tiles = (maxent + 63u) / 64u;
    constexpr unsigned nw = kTtStreamThreads / 64u; // waves of a workgroup
    // workgroups: each stages its rate category's two matrices (2 x 30 KB for 61 states) first, two fit a CU - one round of
    // them (C5, same box: 1 / 2 / 3 / 5 tiles per wave = 1280 / 640 / 448 / 256 workgroups: 101.6 / 106.5 / 99.8 / 98.8 us)
    const unsigned want = 512u;
    unsigned tpw = (unsigned)(((size_t)((tiles + nw - 1u) / nw) * nops * R + want - 1u) / want);
    tpw = std::max(1u, tpw);
    const unsigned nx = (tiles + nw * tpw - 1u) / (nw * tpw);
    const unsigned S = c->gg.S;
    const size_t lds = 2u * (size_t)(S + 1u) * tt_stream_ld_host(S) * sizeof(double);
    const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
    bool scaling = false;
    for (unsigned i = 0; i < nops; ++i) scaling = scaling || pack.ops[i].pscaler != nullptr;
    scaling = scaling && c->gg.scale_mode != 0;
    const unsigned fstride = (maxent + 63u) & ~63u;
    if (scaling && c->mfma_flags.ensure((size_t)kMaxOpsPerLaunch * R * fstride)) return PLLGPU_ENOMEM;
    raise_lds_limit((const void *)k_partials_tt_stream, c->device, lds);
    hipLaunchKernelGGL(k_partials_tt_stream, xcd_grid(nx, nops, R), dim3(kTtStreamThreads), lds, c->stream, pack, c->gg, tm, tpw, c->mfma_flags.p, fstride, nx, nops, c->xcd_order);
    if (scaling)
    {
      dim3 eg((maxent + 255) / 256, nops);
      hipLaunchKernelGGL((k_mfma_scale_epilogue<false>), eg, dim3(256), 0, c->stream, pack, c->gg, c->mfma_flags.p, fstride);
    }
    return 0;
  }
  // aim at two workgroups of four waves on every CU (2048 waves) - four where the small shapes leave room;
  // more work -> more items per wave. (Round 4, C5's tip x tip launch with 1024 / 2048 / 4096 / 8192 / 16384 / 40000
  // waves wanted: 1020 / 1063 / 1076 / 1051 / 1020 / 908 M updates/s for the step - flat around the choice, unlike the
  // 4 x 4 kernels, which gained 6 % from one tile per wave: every workgroup here stages 64 KB of matrices first)
  const unsigned want = NG > 8 ? 2048u : 4096u;
  unsigned ipw = (unsigned)(((size_t)items * nops * R + want - 1) / want);
  ipw = std::max(1u, std::min(ipw, NG > 8 ? ~0u : 8u));
  dim3 grid((items + 4 * ipw - 1) / (4 * ipw), nops, R), block(256);
  // (round 5 ran the plain tip x tip levels - store traffic and nothing else, C5's: 625 MB - in the XCD-aware order of the
  // other store-bound launches as well: 119.7 -> 122.7 us, profiles/README.md; the switch and its code are gone)
  const size_t lds = MfmaGeo<NG>::lds_doubles * sizeof(double);
  const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
  bool scaling = false;
  for (unsigned i = 0; i < nops; ++i) scaling = scaling || pack.ops[i].pscaler != nullptr;
  scaling = scaling && c->gg.scale_mode != 0;
  const unsigned fstride = (maxent + 63u) & ~63u;
  if (scaling && c->mfma_flags.ensure((size_t)kMaxOpsPerLaunch * R * fstride)) return PLLGPU_ENOMEM;
  unsigned char *fb = c->mfma_flags.p;
#define MF_LAUNCH(LT, RT, GA)                                                                                   \
  do                                                                                                            \
  {                                                                                                             \
    raise_lds_limit((const void *)k_partials_mfma<NG, LT, RT, GA>, c->device, lds);                             \
    hipLaunchKernelGGL((k_partials_mfma<NG, LT, RT, GA>), grid, block, lds, c->stream, pack, c->gg, tm, ipw, fb, fstride); \
  } while (0)
  if (kind == 0)
  {
    if (gather) MF_LAUNCH(false, false, true); else MF_LAUNCH(false, false, false);
  }
  else if (kind == 1)
  {
    if (gather) MF_LAUNCH(true, false, true); else MF_LAUNCH(true, false, false);
  }
  else
  {
    if (gather) MF_LAUNCH(true, true, true); else MF_LAUNCH(true, true, false);
  }
#undef MF_LAUNCH
  if (scaling)
  {
    dim3 eg((maxent + 255) / 256, nops);
    if (gather)
      hipLaunchKernelGGL((k_mfma_scale_epilogue<true>), eg, dim3(256), 0, c->stream, pack, c->gg, fb, fstride);
    else
      hipLaunchKernelGGL((k_mfma_scale_epilogue<false>), eg, dim3(256), 0, c->stream, pack, c->gg, fb, fstride);
  }
  return 0;
}

// (tip x tip, tip x tip -> inner x inner) groups on the matrix pipe (kernels_mfma.h: k_partials_mfma_cc)
template <int NG>
static int launch_mfma_cc_t(pllgpu_ctx *c, const FusePack &pack, unsigned ngroups, unsigned entries)
{
  const unsigned R = c->gg.R, S = c->gg.S;
  const unsigned items = (entries + 31) / 32;
  // store-bound: many small workgroups - but each stages its six matrices and two tables (20-50 KB from L2), so not
  // too small: C3 (1563 items x 16 groups x 4 rates) with 8 / 4 / 2 / 1 items per wave and the XCD-aware order: step
  // 0.550 / 0.537 / 0.563 / 0.622 ms on one box (round 4, tools/round4_calls/r4_c3_exp.sh; natural order: 0.555 / - / 0.622 / -);
  // 3 / 4 / 5 / 6 on another: 0.551-0.554 / 0.549-0.558 / 0.550-0.554 / 0.553-0.557 - flat between three and six
  const unsigned want = 4096u;
  unsigned ipw = (unsigned)(((size_t)items * ngroups * R + want - 1) / want);
  ipw = std::max(1u, std::min(ipw, 4u));
  const unsigned nx = (items + 4 * ipw - 1) / (4 * ipw);
  dim3 grid = xcd_grid(nx, ngroups, R), block(256);
  const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
  const unsigned ncodes = c->tip_ncodes;
  const size_t lds = CcGeo<NG>::lds_bytes(ncodes);
  // which cherries are rescaled: per pair of tip codes, every rate's answer (k_cherry_bits) - a table per pair of
  // tip matrices, kept on the device until one of the two is written again
  {
    const unsigned char *before = c->cherry_bits.p;
    if (int rc = c->cherry_bits.ensure((size_t)kCherrySlots * R * ncodes * ncodes)) return rc;
    if (c->cherry_bits.p != before || c->cherry_slot.size() != kCherrySlots)
    {
      c->cherry_slot_of.clear(); // a fresh block: no table survives
      c->cherry_slot.assign(kCherrySlots, pllgpu_ctx::CherrySlot());
    }
  }
  CherryTips stale;
  memset(&stale, 0, sizeof stale);
  unsigned nstale = 0;
  CherrySlots slots;
  memset(&slots, 0, sizeof slots);
  OpPack parents; // the group parents as plain ops: what the scaling epilogue works on
  memset(&parents, 0, sizeof parents);
  bool scaling = false;
  if (c->cherry_slot_of.size() + 2 * ngroups > kCherrySlots)
  {
    // full: start over (launches already in the stream read their tables before any of them is overwritten)
    c->cherry_slot_of.clear();
    c->cherry_slot.assign(kCherrySlots, pllgpu_ctx::CherrySlot());
  }
  for (unsigned i = 0; i < ngroups; ++i)
  {
    for (unsigned ch = 0; ch < 2; ++ch)
    {
      const FOp &f = ch ? pack.g[i].b : pack.g[i].a;
      const unsigned long long li = (unsigned long long)((f.lmat - c->pmat.p) / (ptrdiff_t)c->pm_stride),
                               ri = (unsigned long long)((f.rmat - c->pmat.p) / (ptrdiff_t)c->pm_stride);
      auto ins = c->cherry_slot_of.emplace((li << 32) | ri, (unsigned)c->cherry_slot_of.size());
      const unsigned sl = ins.first->second;
      pllgpu_ctx::CherrySlot &cs = c->cherry_slot[sl];
      slots.s[2 * i + ch] = (unsigned short)sl;
      if (cs.lver != c->pm_version[li] || cs.rver != c->pm_version[ri] || cs.maps != c->maps_epoch + 1ull)
      {
        cs.lver = c->pm_version[li];
        cs.rver = c->pm_version[ri];
        cs.maps = c->maps_epoch + 1ull;
        bool listed = false; // the same pair of matrices twice in one launch: one table
        for (unsigned q = 0; q < nstale; ++q) listed = listed || stale.slot[q] == sl;
        if (!listed)
        {
          stale.lmat[nstale] = f.lmat;
          stale.rmat[nstale] = f.rmat;
          stale.slot[nstale] = (unsigned short)sl;
          ++nstale;
        }
      }
    }
    DevOp &d = parents.ops[i];
    d.parent = pack.g[i].p.parent;
    d.pscaler = pack.g[i].p.pscaler;
    d.lscaler = pack.g[i].a.pscaler;
    d.rscaler = pack.g[i].b.pscaler;
    d.entries = entries;
    scaling = scaling || d.pscaler != nullptr;
  }
  scaling = scaling && c->gg.scale_mode != 0;
  if (nstale)
  {
    size_t bits_lds = ((size_t)2 * S * S + (size_t)2 * ncodes * S) * sizeof(double);
    const unsigned staged = bits_lds <= 144u * 1024u ? 1u : 0u;
    if (!staged) bits_lds = (size_t)2 * ncodes * S * sizeof(double);
    raise_lds_limit((const void *)k_cherry_bits, c->device, bits_lds);
    hipLaunchKernelGGL(k_cherry_bits, dim3(nstale, R), dim3(256), bits_lds, c->stream, stale, c->gg, tm, ncodes, c->cherry_bits.p,
                       nullptr, staged);
  }
  const unsigned fstride = (entries + 63u) & ~63u;
  if (scaling && c->mfma_flags.ensure((size_t)kMaxOpsPerLaunch * R * fstride)) return PLLGPU_ENOMEM;
  raise_lds_limit((const void *)k_partials_mfma_cc<NG>, c->device, lds);
  // parents beyond what the Infinity Cache keeps for the next level: streamed out like the cherries (as the 4x4 groups do)
  const unsigned stream_parent = ((size_t)ngroups * entries * S * R * 8u > c->stream_parent_bytes) ? 1u : 0u;
  hipLaunchKernelGGL((k_partials_mfma_cc<NG>), grid, block, lds, c->stream, pack, c->gg, tm, entries, ipw, c->mfma_flags.p, fstride,
                     c->cherry_bits.p, slots, ncodes, stream_parent, nx, ngroups, c->xcd_order);
  if (scaling)
    hipLaunchKernelGGL((k_mfma_scale_epilogue<false>), dim3((entries + 255) / 256, ngroups), dim3(256), 0, c->stream, parents, c->gg,
                       c->mfma_flags.p, fstride);
  return 0;
}

static int launch_mfma_cc(pllgpu_ctx *c, const FusePack &pack, unsigned ngroups, unsigned entries)
{
  if (c->mfma_ng == 5) return launch_mfma_cc_t<5>(c, pack, ngroups, entries);
  return launch_mfma_cc_t<8>(c, pack, ngroups, entries);
}

static int launch_mfma(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned maxent, unsigned kind, bool gather)
{
  switch (c->mfma_ng)
  {
    case 5: return launch_mfma_t<5>(c, pack, nops, maxent, kind, gather);
    case 8: return launch_mfma_t<8>(c, pack, nops, maxent, kind, gather);
    default: return launch_mfma_t<16>(c, pack, nops, maxent, kind, gather);
  }
}

// 17..20 states, up to four rate categories: every level launch on the matrix pipe (kernels_lean.h)
static int launch_lean(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned maxent, unsigned kind, bool gather)
{
  const unsigned R = c->gg.R;
  const unsigned items = (maxent + 31) / 32;
  // staging the workgroup's matrices costs about as much as two items: several items per workgroup, as long as
  // a few thousand workgroups remain
  unsigned ipb = (unsigned)(((size_t)items * nops) / 4096u);
  ipb = std::max(2u, std::min(ipb, 16u));
  // a tip x tip item is two dozen LDS reads and five stores: staging has to be spread over more of them (C3's 32 cherries:
  // 295 / 240 / 205 / 204 / 254 us with 2 / 8 / 16 / 32 / 64 items per workgroup)
  if (kind == 2 && !gather) ipb = std::max(2u, std::min((unsigned)(((size_t)items * nops) / 1536u), 32u));
  dim3 grid((items + ipb - 1) / ipb, nops), block(64u * R);
  const size_t lds = LeanGeo<5>::lds_bytes(R, gather);
  if (gather)
  {
    raise_lds_limit((const void *)k_partials_lean<5, false, false, true>, c->device, lds);
    raise_lds_limit((const void *)k_partials_lean<5, true, false, true>, c->device, lds);
    raise_lds_limit((const void *)k_partials_lean<5, true, true, true>, c->device, lds);
  }
  const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
  const unsigned ncodes = std::min(c->tip_ncodes, 256u);
#define LEAN_LAUNCH(LT, RT, GA) hipLaunchKernelGGL((k_partials_lean<5, LT, RT, GA>), grid, block, lds, c->stream, pack, c->gg, tm, ipb, ncodes)
  if (kind == 0)
  {
    if (gather) LEAN_LAUNCH(false, false, true); else LEAN_LAUNCH(false, false, false);
  }
  else if (kind == 1)
  {
    if (gather) LEAN_LAUNCH(true, false, true); else LEAN_LAUNCH(true, false, false);
  }
  else
  {
    if (gather) LEAN_LAUNCH(true, true, true); else LEAN_LAUNCH(true, true, false);
  }
#undef LEAN_LAUNCH
  return 0;
}

// the matrix-pipe level kernel takes gathering launches (site repeats) whose inner children are entry-contiguous
// (the rule unless PLL_AMD_NO_GENERIC_AOS=1), states_padded = 20
static bool lean_serves(const pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned kind, bool gather)
{
  if (!gather) return kind == 2; // tip x tip: 204 us against the FMA kernel's 251 for C3's 32 cherries (same box)
  if (c->gg.SP != 20u) return false;
  for (unsigned i = 0; i < nops; ++i)
  {
    const unsigned lay = pack.ops[i].layout;
    if (kind == 0 && !(lay & kAosLeft)) return false;
    if (kind != 2 && !(lay & kAosRight)) return false;
  }
  return true;
}

static int launch_partials(pllgpu_ctx *c, const OpPack &pack, unsigned nops, unsigned maxent, unsigned kind, bool gather)
{
  if (c->dna_fast)
    launch_dna(c, pack, nops, maxent, kind, gather);
  else if (c->use_mfma)
    return launch_mfma(c, pack, nops, maxent, kind, gather);
  else if (c->lean && (kind == 0 || c->tipmap_set) && lean_serves(c, pack, nops, kind, gather))
    return launch_lean(c, pack, nops, maxent, kind, gather);
  else
    switch (c->ich)
    {
      case 4: launch_generic<4>(c, pack, nops, maxent, kind, gather); break;
      case 8: launch_generic<8>(c, pack, nops, maxent, kind, gather); break;
      case 16: launch_generic<16>(c, pack, nops, maxent, kind, gather); break;
      case 20: launch_generic<20>(c, pack, nops, maxent, kind, gather); break;
      default: launch_generic<32>(c, pack, nops, maxent, kind, gather); break;
    }
  return 0;
}

#include "fusion_plan.h"
#include "chain_plan.h"
#include "subtree_plan.h"

// launch the held ops as ordinary updates (they are mutually independent: one level)
static int flush_deferred(pllgpu_ctx *c)
{
  if (int rc = launch_held_chains(c)) return rc;
  std::vector<pllgpu_op_t> ops;
  ops.swap(c->deferred);
  for (unsigned kind = 0; kind < 3; ++kind)
  {
    OpPack pack;
    unsigned nops = 0, maxent = 0;
    for (const pllgpu_op_t &o : ops)
    {
      const unsigned tips = ((o.flags & PLLGPU_OP_LEFT_TIP) ? 1u : 0u) + ((o.flags & PLLGPU_OP_RIGHT_TIP) ? 1u : 0u);
      if (tips != kind) continue;
      if (int rc = resolve_op(c, o, pack.ops[nops])) return rc;
      maxent = std::max(maxent, o.parent_entries);
      ++nops;
    }
    if (!nops) continue;
    if (int rc = launch_partials(c, pack, nops, maxent, kind, false)) return rc;
    ++c->last_launches;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PLLGPU_ERUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

// the level scheduler: plan the list and launch it (every launch through emit(), so that a plan being
// recorded keeps it)
static int plan_and_launch_levels(pllgpu_ctx *c, const pllgpu_op_t *ops, unsigned count)
{
  c->launch_rc = 0;
  std::vector<int> role;
  std::vector<FusedGroup> groups;
  // the groups look their cherries' scaling decisions up in a table over all pairs of tip codes: only for a sane number of codes
  const bool cherry_groups = c->fuse_mfma && c->tipmap_set && c->tip_ncodes <= 32u;
  const bool generic_groups = cherry_groups;
  plan_fusion(c->fuse || generic_groups, c->fuse_cc, c->geo.nodes, ops, count, role, groups);
  if (generic_groups)
  {
    // of the groups the 4x4 planner knows, the other shapes have one: both children cherries
    std::vector<FusedGroup> keep;
    for (const FusedGroup &gq : groups)
      if (gq.lk == CK_FTT && gq.rk == CK_FTT && gq.a >= 0 && gq.b >= 0)
        keep.push_back(gq);
      else
      {
        role[gq.p] = 0;
        if (gq.a >= 0) role[gq.a] = 0;
        if (gq.b >= 0) role[gq.b] = 0;
      }
    groups.swap(keep);
  }
  // site repeats (4x4 kernels): an uncompressed op over two GATHERING inner x inner ops of the level below, all four
  // of their children class-compressed, is evaluated with them (kernels_dna.h: k_partials_dna_gg). Such an op may
  // have been taken as a producer by the op above it; that group gives way (its parent is usually one of the last
  // level's ops, which then wait for the edge evaluation).
  if (c->dna_fast && c->fuse && c->fuse_gg)
  {
    std::vector<int> producer(c->geo.nodes, -1), pl(count, -1), pr(count, -1);
    for (unsigned q = 0; q < count; ++q)
    {
      if (!(ops[q].flags & PLLGPU_OP_LEFT_TIP)) pl[q] = producer[ops[q].left_clv];
      if (!(ops[q].flags & PLLGPU_OP_RIGHT_TIP)) pr[q] = producer[ops[q].right_clv];
      producer[ops[q].parent_clv] = (int)q;
    }
    auto gathering = [&](int x, int pscal, const pllgpu_op_t &P, unsigned L) {
      if (x < 0 || role[x] != 0) return false;
      const pllgpu_op_t &o = ops[x];
      return o.level == L && (o.flags & PLLGPU_OP_GATHER) && !(o.flags & (PLLGPU_OP_LEFT_TIP | PLLGPU_OP_RIGHT_TIP)) &&
             o.parent_entries == P.parent_entries && o.parent_entries == c->geo.sites_alloc && o.parent_scaler == pscal &&
             c->ids[o.left_clv] && c->ids[o.right_clv];
    };
    for (unsigned q = 0; q < count; ++q)
    {
      const pllgpu_op_t &P = ops[q];
      if ((P.flags & (PLLGPU_OP_GATHER | PLLGPU_OP_LEFT_TIP | PLLGPU_OP_RIGHT_TIP)) || P.level == 0 || role[q] == 1) continue;
      const unsigned L = P.level - 1;
      if (P.war_level >= (int)L || pl[q] == pr[q]) continue;
      if (!gathering(pl[q], P.left_scaler, P, L) || !gathering(pr[q], P.right_scaler, P, L)) continue;
      if (role[q] == 2)
      {
        // dissolve the group that took P as a producer
        for (size_t gi = 0; gi < groups.size(); ++gi)
          if ((groups[gi].a == (int)q || groups[gi].b == (int)q) && groups[gi].lk != CK_FCC && groups[gi].rk != CK_FCC)
          {
            role[groups[gi].p] = 0;
            if (groups[gi].a >= 0) role[groups[gi].a] = 0;
            if (groups[gi].b >= 0) role[groups[gi].b] = 0;
            groups.erase(groups.begin() + gi);
            break;
          }
        if (role[q] != 0) continue; // (a cherry-cherry side: leave it)
      }
      FusedGroup gq;
      gq.p = q;
      gq.a = pl[q];
      gq.b = pr[q];
      gq.lk = gq.rk = CK_FGG;
      gq.level = L;
      role[q] = 1;
      role[gq.a] = role[gq.b] = 2;
      groups.push_back(gq);
    }
  }
  // tail fusion: the plain ops of the last level (at most two: the ends of the edge a caller evaluates
  // next) are accepted but not launched yet - role 3
  if (c->defer_tail && count)
  {
    const unsigned top = ops[count - 1].level;
    unsigned n = 0;
    bool ok = true;
    for (unsigned o = 0; o < count && ok; ++o)
      if (ops[o].level == top)
      {
        if (role[o] == 1) continue; // a group parent hoisted away from this level
        ok = role[o] == 0 && !(ops[o].flags & PLLGPU_OP_GATHER) && ops[o].parent_entries == c->geo.sites_alloc && ++n <= 2;
      }
    for (unsigned o = 0; o < count && ok; ++o)
      if (ops[o].level == top && role[o] == 0)
      {
        role[o] = 3;
        c->deferred.push_back(ops[o]);
        c->last_bytes += op_traffic(c, ops[o], true, true);
      }
  }
  // site repeats: every all-tip subtree of up to three ops' depth in one launch, before the levels (role 4)
  {
    unsigned nsub = 0, sub_entries = 0;
    if (int rc = plan_subtrees(c, ops, count, role, nsub, sub_entries)) return rc;
    if (nsub)
    {
      c->last_launches += (nsub + (unsigned)kSubItemsPerLaunch - 1) / (unsigned)kSubItemsPerLaunch;
      emit(c, [c, items = c->sub_build]() {
        if (int rc = upload_subtrees(c, items)) c->launch_rc = rc;
        else launch_subtrees(c, (unsigned)items.size());
      });
      if (c->launch_rc) return c->launch_rc;
    }
  }
  size_t gi_sorted = 0;
  if (!groups.empty())
    sort_groups_by_level(groups);
  unsigned i = 0;
  unsigned level = 0;
  const unsigned last_level = count ? ops[count - 1].level : 0;
  for (level = 0; level <= last_level; ++level)
  {
    // [i, j) = the ops of this dependency level
    unsigned j = i;
    while (j < count && ops[j].level == level) ++j;
    // plain ops: one launch group per (child kinds, gather) in packs of kMaxOpsPerLaunch
    for (unsigned kind = 0; kind < 3; ++kind)
      for (unsigned ga = 0; ga < 2; ++ga)
      {
        OpPack pack;
        unsigned nops = 0, maxent = 0;
        int lrc = 0;
        auto flush = [&]() {
          if (!nops) return;
          const bool gather = ga != 0;
          emit(c, [c, pack, nops, maxent, kind, gather]() {
            if (int rc = launch_partials(c, pack, nops, maxent, kind, gather)) c->launch_rc = rc;
          });
          if (c->launch_rc) lrc = c->launch_rc;
          ++c->last_launches;
          nops = 0;
          maxent = 0;
        };
        for (unsigned o = i; o < j; ++o)
        {
          const unsigned f = ops[o].flags;
          const unsigned tips = ((f & PLLGPU_OP_LEFT_TIP) ? 1u : 0u) + ((f & PLLGPU_OP_RIGHT_TIP) ? 1u : 0u);
          if ((f & PLLGPU_OP_RIGHT_TIP) && !(f & PLLGPU_OP_LEFT_TIP))
            return fail(PLLGPU_EINVAL, "tip-inner operations must carry the tip as the left child");
          if (role[o] || tips != kind || ((f & PLLGPU_OP_GATHER) ? 1u : 0u) != ga) continue;
          if (ops[o].parent_entries == 0) continue;
          if (int rc = resolve_op(c, ops[o], pack.ops[nops])) return rc;
          c->last_bytes += op_traffic(c, ops[o], true, true);
          maxent = std::max(maxent, ops[o].parent_entries);
          if (++nops == (unsigned)kMaxOpsPerLaunch) flush();
        }
        flush();
        if (lrc) return lrc;
      }
    // fused groups executing at this level, one launch per pair of child kinds
    const size_t g0 = gi_sorted;
    while (gi_sorted < groups.size() && groups[gi_sorted].level == level) ++gi_sorted;
    // cherry-cherry groups (kind CK_FCC on at least one side) have their own descriptor and kernel
    {
      std::vector<CCLaunch> ccl;
      if (int rc = build_cc_launches(c, ops, groups, g0, gi_sorted, ccl)) return rc;
      for (const CCLaunch &l : ccl)
      {
        emit(c, [c, l]() {
          if (int rc = launch_cc(c, l.pack, l.n, l.entries, l.lk, CK_FCC)) c->launch_rc = rc;
        });
        if (c->launch_rc) return c->launch_rc;
        ++c->last_launches;
      }
    }
    // groups over two gathering producers (site repeats)
    {
      GGPack gp;
      unsigned n = 0, entries = 0;
      auto flushg = [&]() -> int {
        if (!n) return 0;
        emit(c, [c, gp, n, entries]() { launch_gg(c, gp, n, entries); });
        ++c->last_launches;
        n = 0;
        return c->launch_rc;
      };
      for (size_t gi = g0; gi < gi_sorted; ++gi)
      {
        const FusedGroup &gq = groups[gi];
        if (gq.lk != CK_FGG) continue;
        const pllgpu_op_t &P = ops[gq.p];
        if (P.parent_entries == 0) continue;
        if (n && P.parent_entries != entries)
          if (int rc = flushg()) return rc;
        entries = P.parent_entries;
        GGroup &gg = gp.g[n];
        memset(&gg, 0, sizeof gg);
        DevOp d;
        if (int rc = resolve_op(c, ops[gq.a], gg.a)) return rc;
        if (int rc = resolve_op(c, ops[gq.b], gg.b)) return rc;
        if (int rc = resolve_op(c, P, d)) return rc;
        to_fop(d, gg.p);
        c->last_bytes += op_traffic(c, ops[gq.a], true, true) + op_traffic(c, ops[gq.b], true, true) + op_traffic(c, P, false, false);
        if (++n == (unsigned)kMaxGGroups)
          if (int rc = flushg()) return rc;
      }
      if (int rc = flushg()) return rc;
    }
    for (int lk = 0; lk <= CK_FII; ++lk)
      for (int rk = lk; rk <= CK_FII; ++rk)
      {
        FusePack pack;
        unsigned n = 0, entries = 0;
        auto flush = [&]() -> int {
          if (!n) return 0;
          if (!c->dna_fast)
            emit(c, [c, pack, n, entries]() {
              if (int rc = launch_mfma_cc(c, pack, n, entries)) c->launch_rc = rc;
            });
          else
            emit(c, [c, pack, n, entries, lk, rk]() {
              if (int rc = launch_fused(c, pack, n, entries, lk, rk)) c->launch_rc = rc;
            });
          if (c->launch_rc) return c->launch_rc;
          ++c->last_launches;
          n = 0;
          return 0;
        };
        for (size_t gi = g0; gi < gi_sorted; ++gi)
        {
          FusedGroup g = groups[gi];
          if (g.lk == CK_FCC || g.rk == CK_FCC || g.lk == CK_FGG) continue; // launched above
          const bool swap = g.lk > g.rk; // canonical order: the "smaller" kind on the left
          if ((swap ? g.rk : g.lk) != lk || (swap ? g.lk : g.rk) != rk) continue;
          const pllgpu_op_t &P = ops[g.p];
          if (P.parent_entries == 0) continue;
          if (n && P.parent_entries != entries)
            if (int rc = flush()) return rc;
          entries = P.parent_entries;
          DevOp dp, da, db;
          FGroup &fg = pack.g[n];
          memset(&fg, 0, sizeof fg);
          // producers first: their parent buffers must exist before P's children are resolved
          if (g.a >= 0)
          {
            if (int rc = resolve_op(c, ops[g.a], da)) return rc;
            c->last_bytes += op_traffic(c, ops[g.a], true, true);
          }
          if (g.b >= 0)
          {
            if (int rc = resolve_op(c, ops[g.b], db)) return rc;
            c->last_bytes += op_traffic(c, ops[g.b], true, true);
          }
          if (int rc = resolve_op(c, P, dp)) return rc;
          c->last_bytes += op_traffic(c, P, g.a < 0, g.b < 0);
          to_fop(dp, fg.p);
          if (g.a >= 0) to_fop(da, fg.a);
          if (g.b >= 0) to_fop(db, fg.b);
          if (swap)
          {
            std::swap(fg.p.left, fg.p.right);
            std::swap(fg.p.ltip, fg.p.rtip);
            std::swap(fg.p.lscaler, fg.p.rscaler);
            std::swap(fg.p.lmat, fg.p.rmat);
            std::swap(fg.a, fg.b);
          }
          if (++n == (unsigned)kMaxGroups)
            if (int rc = flush()) return rc;
        }
        if (int rc = flush()) return rc;
      }
    i = j;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PLLGPU_ERUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int pllgpu_update_partials(pllgpu_ctx_t *c, const pllgpu_op_t *ops, unsigned count)
{
  CHECK_CTX(c);
  c->last_launches = 0;
  c->last_bytes = 0.0;
  {
    bool used = false;
    if (int rc = try_chain_plan(c, ops, count, used)) return rc;
    if (used) return 0;
    c->last_bytes = 0.0;
  }
  if (c->plan_cache)
  {
    const unsigned long long epoch = c->alloc_epoch;
    for (LevelPlan *lp : c->level_plans)
      if (lp->alloc_epoch == epoch && lp->maps_epoch == c->maps_epoch && lp->key.size() == count &&
          memcmp(lp->key.data(), ops, count * sizeof(pllgpu_op_t)) == 0)
      {
        // the same list, nothing it points at has moved: the same launches
        for (const auto &a : lp->aos) c->clv_aos[a.first] = a.second;
        if (lp->any_aos) c->any_aos = true;
        c->deferred = lp->deferred;
        c->launch_rc = 0;
        for (const auto &fn : lp->launches) fn();
        ++c->plan_replays;
        c->last_launches = lp->nlaunches;
        c->last_bytes = lp->bytes;
        lp->used = ++c->plan_stamp;
        if (c->launch_rc) return c->launch_rc;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(PLLGPU_ERUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
        return 0;
      }
  }
  LevelPlan *lp = c->plan_cache ? new LevelPlan() : nullptr;
  c->recording = lp;
  const int rc = plan_and_launch_levels(c, ops, count);
  c->recording = nullptr;
  if (!lp) return rc;
  if (rc)
  {
    delete lp;
    return rc;
  }
  lp->key.assign(ops, ops + count);
  lp->alloc_epoch = c->alloc_epoch; // after the planning: it may have allocated
  lp->maps_epoch = c->maps_epoch;
  lp->deferred = c->deferred;
  for (unsigned i = 0; i < count; ++i) lp->aos.emplace_back(ops[i].parent_clv, c->clv_aos[ops[i].parent_clv]);
  lp->any_aos = c->any_aos;
  lp->nlaunches = c->last_launches;
  lp->bytes = c->last_bytes;
  lp->used = ++c->plan_stamp;
  if (c->level_plans.size() >= kLevelPlans)
  {
    size_t victim = 0;
    for (size_t i = 1; i < c->level_plans.size(); ++i)
      if (c->level_plans[i]->used < c->level_plans[victim]->used) victim = i;
    delete c->level_plans[victim];
    c->level_plans[victim] = lp;
  }
  else
    c->level_plans.push_back(lp);
  return 0;
}

extern "C" double pllgpu_last_algorithmic_bytes(const pllgpu_ctx_t *c) { return c ? c->last_bytes : 0.0; }

// ---- log-likelihood ----------------------------------------------------------------------------
template <int ICH>
static void launch_edge_generic(pllgpu_ctx *c, const DevEdge &e, unsigned blocks, unsigned tpw, bool ctip, bool gather)
{
  const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
  const unsigned threads = 64u * std::min(c->gg.R, 4u); // one wave per rate category of the tile, up to 4
#define EG(CT, GA) hipLaunchKernelGGL((k_edge_tiled<ICH, CT, GA>), dim3(blocks), dim3(threads), 0, c->stream, e, c->gg, tm, tpw)
  if (ctip)
  {
    if (gather) EG(true, true); else EG(true, false);
  }
  else
  {
    if (gather) EG(false, true); else EG(false, false);
  }
#undef EG
}

// the tail-fused evaluation: kinds of the two ends and the descriptors of the held ops
struct TailCall
{
  FGroup g;
  int kp, kc;
};

static int launch_edge_tail(pllgpu_ctx *c, const DevEdge &e, const TailCall &t, unsigned blocks, unsigned tpw)
{
#define TZ(A, B)                                                                                                      \
  if (t.kp == A && t.kc == B)                                                                                         \
  {                                                                                                                   \
    hipLaunchKernelGGL((k_edge_dna_tail<A, B>), dim3(blocks), dim3(256), 0, c->stream, e, t.g, c->gg.scale_mode, tpw); \
    return 0;                                                                                                         \
  }
  TZ(CK_INNER, CK_FTT) TZ(CK_INNER, CK_FTI) TZ(CK_INNER, CK_FII)
  TZ(CK_FTT, CK_INNER) TZ(CK_FTT, CK_TIP) TZ(CK_FTT, CK_FTT) TZ(CK_FTT, CK_FTI) TZ(CK_FTT, CK_FII)
  TZ(CK_FTI, CK_INNER) TZ(CK_FTI, CK_TIP) TZ(CK_FTI, CK_FTT) TZ(CK_FTI, CK_FTI) TZ(CK_FTI, CK_FII)
  TZ(CK_FII, CK_INNER) TZ(CK_FII, CK_TIP) TZ(CK_FII, CK_FTT) TZ(CK_FII, CK_FTI) TZ(CK_FII, CK_FII)
#undef TZ
  return fail(PLLGPU_EINVAL, "no tail kernel for end kinds (%d, %d)", t.kp, t.kc);
}

// chain tail (kernels_dna.h: k_edge_dna_chain): the two ends of the edge as chains - held ones, or chains of
// no steps around a CLV / tip that is in HBM
struct ChainTailCall
{
  ChainHead hp, hc;
  unsigned variant;
  bool in_kernarg;
  ChainPack pack; // in_kernarg: heads[0] = hp, heads[1] = hc and their steps
};

static void launch_edge_chain(pllgpu_ctx *c, const DevEdge &e, const ChainTailCall &t)
{
  const ChainPlan &pl = *c->plan;
  dim3 grid((pl.entries + 63) / 64), block(256);
  if (t.in_kernarg)
  {
#define EC(SMV, C0, S1, C1) hipLaunchKernelGGL((k_edge_dna_chain_pack<SMV, C0, S1, C1>), grid, block, 0, c->stream, e, t.pack, pl.entries)
#define EC_V(SMV)                              \
  switch (t.variant)                           \
  {                                            \
  case 0: EC(SMV, false, false, false); break; \
  case 1: EC(SMV, false, true, false); break;  \
  case 2: EC(SMV, true, false, false); break;  \
  default: EC(SMV, true, true, true); break;   \
  }
    if (c->gg.scale_mode == 2)
    {
      EC_V(2)
    }
    else
    {
      EC_V(1)
    }
#undef EC_V
#undef EC
  }
  else
  {
    const unsigned char *base = c->chain_dev.p;
    const size_t heads_bytes = pl.heads.size() * sizeof(ChainHead), loads_bytes = pl.loads.size() * sizeof(ChainStepLoad);
    const ChainStepLoad *lp = reinterpret_cast<const ChainStepLoad *>(base + heads_bytes);
    const ChainStepOp *op = reinterpret_cast<const ChainStepOp *>(base + heads_bytes + loads_bytes);
#define EC(SMV, C0, S1, C1) hipLaunchKernelGGL((k_edge_dna_chain<SMV, C0, S1, C1>), grid, block, 0, c->stream, e, t.hp, t.hc, lp, op, pl.entries)
#define EC_V(SMV)                              \
  switch (t.variant)                           \
  {                                            \
  case 0: EC(SMV, false, false, false); break; \
  case 1: EC(SMV, false, true, false); break;  \
  case 2: EC(SMV, true, false, false); break;  \
  default: EC(SMV, true, true, true); break;   \
  }
    if (c->gg.scale_mode == 2)
    {
      EC_V(2)
    }
    else
    {
      EC_V(1)
    }
#undef EC_V
#undef EC
  }
}

static int run_lnl(pllgpu_ctx *c, DevEdge &e, bool ctip, bool gather, const unsigned *freqs_indices,
                   double *persite_host, double *lnl_out, double *device_result = nullptr, const TailCall *tail = nullptr,
                   const ChainTailCall *ctail = nullptr)
{
  const pllgpu_geometry_t &g = c->geo;
  for (unsigned k = 0; k < g.rate_cats; ++k)
  {
    if (freqs_indices[k] >= g.rate_matrices) return fail(PLLGPU_EINVAL, "freqs_indices[%u] = %u out of range", k, freqs_indices[k]);
    e.fidx[k] = (unsigned char)freqs_indices[k];
  }
  e.freqs = c->freqs.p;
  e.rate_weights = c->rate_weights.p;
  e.prop_invar = c->prop_invar.p;
  e.pattern_weights = c->pattern_weights.p;
  e.invariant = c->invariant_set ? c->invariant.p : nullptr;
  e.persite = persite_host ? c->persite.p : nullptr;
  e.block_sums = c->block_sums.p;
  e.counter = c->counter.p;
  e.result = device_result ? device_result : c->result_dev;
  if (c->seq_override != 0.0)
    e.sequence = c->seq_override; // a collective's step number: only meaningful in device_result
  else
  {
    c->seq += 1.0;
    e.sequence = c->seq;
  }
  unsigned long long seq_bits;
  memcpy(&seq_bits, &e.sequence, sizeof seq_bits);
  e.sites = g.sites;
  e.per_rate = g.per_rate_scalers ? 1 : 0;
  e.fenced = c->fenced;

  const unsigned tiles = (g.sites + 63) / 64;
  const unsigned max_blocks = 1024;
  unsigned blocks, tpw;
  if (c->use_mfma && c->mfma_ng == 16 && !e.is_root)
  {
    // 33..64 states: P x on the matrix pipe (kernels_mfma.h: k_edge_mfma), one rate category per workgroup
    const unsigned items = (g.sites + 31) / 32, R = c->gg.R;
    // a workgroup's life is a chain of round trips (matrix, child, parent, ticket, partials, block sum: ~30 us at C5's
    // size) around 3 us of MFMAs per item, and two workgroups fit a CU: ONE round of at most 512 workgroups - C5's 625
    // items as 628 workgroups of one item per wave took two rounds, 67 us; as 316 of two items per wave 3x us
    const unsigned max_ib = std::max(1u, 512u / R);
    const unsigned ipw = std::max(1u, (items + 4u * max_ib - 1) / (4u * max_ib));
    const unsigned nib = (items + 4u * ipw - 1) / (4u * ipw);
    blocks = nib * R;
    const unsigned pstride = (g.sites + 63u) & ~63u;
    if (c->block_sums.ensure(std::max<size_t>(4096, blocks)) || c->edge_partials.ensure((size_t)R * 4u * pstride)) return PLLGPU_ENOMEM;
    if (c->edge_tickets.cap < nib)
    {
      if (c->edge_tickets.ensure(std::max(1024u, nib))) return PLLGPU_ENOMEM;
      HIP_TRY(hipMemsetAsync(c->edge_tickets.p, 0, c->edge_tickets.cap * sizeof(unsigned), c->stream));
    }
    e.block_sums = c->block_sums.p;
    const size_t lds = (kFragArray + 64) * sizeof(double);
    const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
#define EM(CT, GA) hipLaunchKernelGGL((k_edge_mfma<CT, GA>), dim3(blocks), dim3(256), lds, c->stream, e, c->gg, tm, ipw, c->edge_partials.p, pstride, c->edge_tickets.p)
    if (ctip)
    {
      if (gather) EM(true, true); else EM(true, false);
    }
    else
    {
      if (gather) EM(false, true); else EM(false, false);
    }
#undef EM
  }
  else if (c->dna_fast && ctail)
  {
    launch_edge_chain(c, e, *ctail);
    ++c->last_launches;
  }
  else if (c->dna_fast && tail)
  {
    tpw = (tiles + 4 * max_blocks - 1) / (4 * max_blocks);
    blocks = (tiles + 4 * tpw - 1) / (4 * tpw);
    if (int rc = launch_edge_tail(c, e, *tail, blocks, tpw)) return rc;
    ++c->last_launches;
  }
  else if (c->dna_fast)
  {
    // 4 waves = 4 tiles per workgroup, tpw consecutive tiles per wave
    tpw = (tiles + 4 * max_blocks - 1) / (4 * max_blocks);
    blocks = (tiles + 4 * tpw - 1) / (4 * tpw);
#define ED(CT, GA) hipLaunchKernelGGL((k_edge_dna<CT, GA>), dim3(blocks), dim3(256), 0, c->stream, e, tpw)
    if (ctip)
    {
      if (gather) ED(true, true); else ED(true, false);
    }
    else
    {
      if (gather) ED(false, true); else ED(false, false);
    }
#undef ED
  }
  else
  {
    // one tile per workgroup pass (the waves split its rate categories), tpw tiles per workgroup
    tpw = (tiles + max_blocks - 1) / max_blocks;
    blocks = (tiles + tpw - 1) / tpw;
    switch (c->ich)
    {
      case 4: launch_edge_generic<4>(c, e, blocks, tpw, ctip, gather); break;
      case 8: launch_edge_generic<8>(c, e, blocks, tpw, ctip, gather); break;
      case 16: launch_edge_generic<16>(c, e, blocks, tpw, ctip, gather); break;
      case 20: launch_edge_generic<20>(c, e, blocks, tpw, ctip, gather); break;
      default: launch_edge_generic<32>(c, e, blocks, tpw, ctip, gather); break;
    }
  }
  HIP_TRY(hipGetLastError());
  if (device_result) return 0; // asynchronous: the value stays on the device
  if (persite_host)
  {
    HIP_TRY(hipMemcpyAsync(persite_host, c->persite.p, g.sites * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  else
  {
    // the last workgroup stores {lnL, sequence} into mapped host memory: poll the sequence word for
    // a bounded time (a stream synchronise costs ~15 us of wake-up latency on a ~5 us kernel), then
    // fall back to the synchronise so a failed launch cannot hang the caller
    volatile double *res = c->result_host;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n((const unsigned long long *)&res[1], __ATOMIC_ACQUIRE) != seq_bits)
    {
      if ((++spins & 1023u) == 0 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
      {
        HIP_TRY(hipStreamSynchronize(c->stream));
        break;
      }
    }
  }
  *lnl_out = c->result_host[0];
  return 0;
}

static int scaler_ptr(pllgpu_ctx *c, int idx, const unsigned *&out)
{
  out = nullptr;
  if (idx < 0) return 0;
  if ((unsigned)idx >= c->geo.scale_buffers || !c->scaler[idx].p)
    return fail(PLLGPU_EINVAL, "scale buffer %d unavailable on the device", idx);
  out = c->scaler[idx].p;
  return 0;
}

// Work that pllgpu_edge_loglikelihood has taken out of the held state (chain heads, deferred ops) in order
// to evaluate it inside the log-likelihood kernel. If the call fails before that kernel is launched the
// host layer has already marked those CLVs and scalers as device-side, so they must still be computed:
// the guard launches them as ordinary updates on every exit path that does not disarm it.
struct ClaimedWork
{
  pllgpu_ctx *c;
  int heads[2] = {-1, -1};
  std::vector<pllgpu_op_t> ops;
  bool armed = false;
  explicit ClaimedWork(pllgpu_ctx *ctx) : c(ctx) {}
  void disarm() { armed = false; }
  ~ClaimedWork()
  {
    if (!armed) return;
    char keep[sizeof g_err];
    memcpy(keep, g_err, sizeof keep); // the error that brought us here is the one to report
    if (c->plan)
      for (int h : heads)
        if (h >= 0)
        {
          launch_chain_heads(c, *c->plan, (unsigned)h, 1, c->plan->head_variant[h]);
          ++c->last_launches;
        }
    if (!ops.empty())
    {
      c->deferred.insert(c->deferred.end(), ops.begin(), ops.end());
      (void)flush_deferred(c);
    }
    (void)hipGetLastError();
    memcpy(g_err, keep, sizeof keep);
  }
};

extern "C" int pllgpu_edge_loglikelihood(pllgpu_ctx_t *c, const pllgpu_edge_t *ed, double *persite_host, double *lnl_out)
{
  CHECK_CTX_KEEP(c);
  const pllgpu_geometry_t &g = c->geo;
  // everything that can be checked without touching the held state comes first
  if (ed->parent_clv >= g.nodes || ed->child_clv >= g.nodes || ed->matrix >= g.prob_matrices)
    return fail(PLLGPU_EINVAL, "edge references an index out of range");
  if (ed->parent_scaler >= (int)g.scale_buffers || ed->child_scaler >= (int)g.scale_buffers)
    return fail(PLLGPU_EINVAL, "edge references a scale buffer out of range");
  if (ed->device_result && persite_host) return fail(PLLGPU_EINVAL, "per-site values are not available from an asynchronous evaluation");
  if (ed->sequence != 0.0 && !ed->device_result) return fail(PLLGPU_EINVAL, "a caller-numbered evaluation needs device_result");
  struct SeqScope
  {
    pllgpu_ctx *c;
    ~SeqScope() { c->seq_override = 0.0; }
  } seq_scope_{c};
  c->seq_override = ed->sequence;
  if (ed->child_is_tip && (ed->child_clv >= g.tips || !c->tipchars[ed->child_clv].p))
    return fail(PLLGPU_EINVAL, "tip %u has no codes on the device", ed->child_clv);
  for (unsigned k = 0; k < g.rate_cats; ++k)
    if (ed->freqs_indices[k] >= g.rate_matrices) return fail(PLLGPU_EINVAL, "freqs_indices[%u] = %u out of range", k, ed->freqs_indices[k]);
  ClaimedWork claimed(c);
  // held ops (tail fusion): if they produce an end of THIS edge they are evaluated inside the lnL
  // kernel; whatever else is held goes out as ordinary updates first
  TailCall tail;
  bool use_tail = false;
  int held_p = -1, held_c = -1; // heads of the held chains that end in this edge's parent / child end
  if (c->chain_held)
  {
    const ChainPlan &pl = *c->plan;
    if (c->dna_fast && !ed->gather)
      for (size_t i = pl.held_from; i < pl.stages.size(); ++i)
        for (unsigned h = pl.stages[i].first_head; h < pl.stages[i].first_head + pl.stages[i].nchains; ++h)
        {
          if (pl.head_top_clv[h] == ed->parent_clv && pl.head_top_scaler[h] == ed->parent_scaler) held_p = (int)h;
          else if (!ed->child_is_tip && pl.head_top_clv[h] == ed->child_clv && pl.head_top_scaler[h] == ed->child_scaler) held_c = (int)h;
        }
    if (held_p < 0 && held_c < 0)
    {
      if (int rc = launch_held_chains(c)) return rc;
    }
    else
    {
      // a tail that would not fit its kernarg pack (try_chain_plan keeps such plans in memory: cannot happen
      // there; defensive): nothing is claimed, both chains go out as ordinary launches
      if (pl.in_kernarg)
      {
        const unsigned np = held_p >= 0 ? pl.heads[held_p].nsteps + 1 : 1u, nc = held_c >= 0 ? pl.heads[held_c].nsteps + 1 : 1u;
        if (np + nc > (unsigned)kChainPackSteps) held_p = held_c = -1;
      }
      if (held_p < 0 && held_c < 0)
      {
        if (int rc = launch_held_chains(c)) return rc;
      }
      else
      {
        if (int rc = c->block_sums.ensure((pl.entries + 63) / 64)) return rc; // before anything is claimed
        // whatever else is held goes out as an ordinary chain launch
        c->chain_held = false;
        claimed.heads[0] = held_p;
        claimed.heads[1] = held_c;
        claimed.armed = true;
        for (size_t i = pl.held_from; i < pl.stages.size(); ++i)
          for (unsigned h = pl.stages[i].first_head; h < pl.stages[i].first_head + pl.stages[i].nchains; ++h)
            if ((int)h != held_p && (int)h != held_c)
            {
              launch_chain_heads(c, pl, h, 1, pl.stages[i].variant);
              ++c->last_launches;
            }
      }
    }
  }
  if (!c->deferred.empty())
  {
    int ia = -1, ib = -1;
    if (c->dna_fast && !ed->gather)
      for (size_t i = 0; i < c->deferred.size(); ++i)
      {
        const pllgpu_op_t &o = c->deferred[i];
        if (o.parent_clv == ed->parent_clv && o.parent_scaler == ed->parent_scaler) ia = (int)i;
        else if (!ed->child_is_tip && o.parent_clv == ed->child_clv && o.parent_scaler == ed->child_scaler) ib = (int)i;
      }
    if (ia < 0 && ib < 0)
    {
      if (int rc = flush_deferred(c)) return rc;
    }
    else
    {
      std::vector<pllgpu_op_t> held;
      held.swap(c->deferred);
      for (size_t i = 0; i < held.size(); ++i)
        if ((int)i != ia && (int)i != ib) c->deferred.push_back(held[i]);
      if (ia >= 0) claimed.ops.push_back(held[ia]);
      if (ib >= 0) claimed.ops.push_back(held[ib]);
      claimed.armed = true;
      if (!c->deferred.empty())
        if (int rc = flush_deferred(c)) return rc;
      memset(&tail.g, 0, sizeof tail.g);
      DevOp d;
      if (ia >= 0)
      {
        if (int rc = resolve_op(c, held[ia], d)) return rc;
        to_fop(d, tail.g.a);
      }
      if (ib >= 0)
      {
        if (int rc = resolve_op(c, held[ib], d)) return rc;
        to_fop(d, tail.g.b);
      }
      tail.kp = ia >= 0 ? child_kind(held[ia]) : CK_INNER;
      tail.kc = ib >= 0 ? child_kind(held[ib]) : (ed->child_is_tip ? CK_TIP : CK_INNER);
      use_tail = true;
    }
  }
  DevEdge e;
  memset(&e, 0, sizeof e);
  if (!c->clv[ed->parent_clv].p) return fail(PLLGPU_EINVAL, "CLV %u was never computed or uploaded", ed->parent_clv);
  e.parent = c->clv[ed->parent_clv].p;
  if (ed->child_is_tip)
  {
    if (ed->child_clv >= g.tips || !c->tipchars[ed->child_clv].p) return fail(PLLGPU_EINVAL, "tip %u has no codes on the device", ed->child_clv);
    e.ctip = c->tipchars[ed->child_clv].p;
  }
  else
  {
    if (!c->clv[ed->child_clv].p) return fail(PLLGPU_EINVAL, "CLV %u was never computed or uploaded", ed->child_clv);
    e.child = c->clv[ed->child_clv].p;
  }
  if (int rc = scaler_ptr(c, ed->parent_scaler, e.pscaler)) return rc;
  if (int rc = scaler_ptr(c, ed->child_scaler, e.cscaler)) return rc;
  e.mat = c->pmat.p + (size_t)ed->matrix * c->pm_stride;
  e.layout = (c->clv_aos[ed->parent_clv] ? kAosParent : 0u) | ((!ed->child_is_tip && c->clv_aos[ed->child_clv]) ? kAosLeft : 0u);
  if (e.layout && !ed->gather) return fail(PLLGPU_EINVAL, "a class-compressed CLV met an evaluation without the gather flag");
  if (ed->gather)
  {
    e.psid = c->ids[ed->parent_clv] ? wide_map(c, ed->parent_clv) : nullptr;
    e.csid = c->ids[ed->child_clv] ? wide_map(c, ed->child_clv) : nullptr;
  }
  e.is_root = 0;
  if (held_p >= 0 || held_c >= 0)
  {
    const ChainPlan &pl = *c->plan;
    ChainTailCall ct;
    memset(&ct.hp, 0, sizeof ct.hp);
    memset(&ct.hc, 0, sizeof ct.hc);
    const unsigned clv_bytes = (unsigned)(clv_elems(c, pl.entries) * sizeof(double));
    const unsigned sc_bytes = pl.entries * (c->gg.scale_mode == 2 ? 16u : 4u);
    const unsigned any_end = pl.heads[0].first + pl.heads[0].nsteps; // a CS_END step: what a chain of no steps "fetches"
    if (held_p >= 0)
      ct.hp = pl.heads[held_p];
    else
    {
      ct.hp.acc0.data = e.parent;
      ct.hp.acc0.scaler = e.pscaler;
      ct.hp.bacc.clv = clv_bytes;
      ct.hp.bacc.aux = e.pscaler ? sc_bytes : 0u;
      ct.hp.first = any_end;
    }
    if (held_c >= 0)
      ct.hc = pl.heads[held_c];
    else if (ed->child_is_tip)
    {
      ct.hc.acc0.data = e.ctip;
      ct.hc.bacc.aux = (pl.entries + 3u) & ~3u;
      ct.hc.acc_tip = 1u;
      ct.hc.first = any_end;
    }
    else
    {
      ct.hc.acc0.data = e.child;
      ct.hc.acc0.scaler = e.cscaler;
      ct.hc.bacc.clv = clv_bytes;
      ct.hc.bacc.aux = e.cscaler ? sc_bytes : 0u;
      ct.hc.first = any_end;
    }
    const int vp = held_p >= 0 ? (int)pl.head_variant[held_p] : -1, vc = held_c >= 0 ? (int)pl.head_variant[held_c] : -1;
    ct.variant = (unsigned)((vp < 0) ? vc : (vc < 0) ? vp : (vp == vc ? vp : 3));
    ct.in_kernarg = pl.in_kernarg;
    if (ct.in_kernarg)
    {
      memset(&ct.pack, 0, sizeof ct.pack);
      unsigned ns = 0;
      ChainHead *hh[2] = {&ct.hp, &ct.hc};
      const int held[2] = {held_p, held_c};
      for (int k = 0; k < 2; ++k)
      {
        const unsigned nst = held[k] >= 0 ? hh[k]->nsteps : 0u;
        if (ns + nst + 1 > (unsigned)kChainPackSteps) return fail(PLLGPU_ERUNTIME, "chain tail does not fit its descriptor pack");
        if (held[k] >= 0)
        {
          memcpy(&ct.pack.loads[ns], &pl.loads[hh[k]->first], (nst + 1) * sizeof(ChainStepLoad));
          memcpy(&ct.pack.ops[ns], &pl.sops[hh[k]->first], (nst + 1) * sizeof(ChainStepOp));
        }
        else
          ct.pack.loads[ns].flags = CS_END;
        hh[k]->first = ns;
        ns += nst + 1;
        ct.pack.heads[k] = *hh[k];
      }
    }
    const int rc = run_lnl(c, e, ed->child_is_tip != 0, false, ed->freqs_indices, persite_host, lnl_out, ed->device_result, nullptr, &ct);
    if (rc == 0) claimed.disarm();
    return rc;
  }
  if (use_tail)
  {
    // memory-side descriptors of the ends that were NOT held
    tail.g.p.left = e.parent;
    tail.g.p.lscaler = e.pscaler;
    tail.g.p.right = e.child;
    tail.g.p.rtip = e.ctip;
    tail.g.p.rscaler = e.cscaler;
    const int rc = run_lnl(c, e, ed->child_is_tip != 0, false, ed->freqs_indices, persite_host, lnl_out, ed->device_result, &tail);
    if (rc == 0) claimed.disarm();
    return rc;
  }
  return run_lnl(c, e, ed->child_is_tip != 0, ed->gather != 0, ed->freqs_indices, persite_host, lnl_out, ed->device_result);
}

extern "C" int pllgpu_root_loglikelihood(pllgpu_ctx_t *c, unsigned clv, int scaler, unsigned gather,
                                         const unsigned *freqs_indices, double *persite_host, double *lnl_out)
{
  CHECK_CTX(c);
  if (clv >= c->geo.nodes || !c->clv[clv].p) return fail(PLLGPU_EINVAL, "CLV %u unavailable on the device", clv);
  DevEdge e;
  memset(&e, 0, sizeof e);
  e.parent = c->clv[clv].p;
  if (int rc = scaler_ptr(c, scaler, e.pscaler)) return rc;
  if (gather) e.psid = c->ids[clv] ? wide_map(c, clv) : nullptr;
  e.layout = c->clv_aos[clv] ? kAosParent : 0u;
  if (e.layout && !gather) return fail(PLLGPU_EINVAL, "a class-compressed CLV met an evaluation without the gather flag");
  e.is_root = 1;
  return run_lnl(c, e, false, gather != 0, freqs_indices, persite_host, lnl_out);
}

// ---- the flat seam's tip-tip pair: the caller's lookup table in the reference's layout ---------------------
static unsigned lookup_shift(unsigned states, unsigned ncodes)
{
  if (states == 4) return 4; // 16 j + k (src/core_partials.c:1013-1071)
  unsigned sh = 0;
  while ((1u << sh) < ncodes) ++sh; // ceil(log2(maxstates)), :1152
  return sh;
}

extern "C" int pllgpu_create_lookup(pllgpu_ctx_t *c, double *lookup_host, const double *left_host, const double *right_host,
                                    const unsigned long long *tipmap_host, unsigned ncodes)
{
  CHECK_CTX(c);
  const unsigned S = c->gg.S, SP = c->gg.SP, R = c->gg.R, span = R * SP, shift = lookup_shift(S, ncodes);
  if (S == 4) ncodes = 16;
  if (!ncodes || ncodes > 256) return fail(PLLGPU_EINVAL, "lookup table over %u tip codes", ncodes);
  const size_t entries = (size_t)((ncodes - 1u) << shift) + ncodes, tab = entries * span, mat = (size_t)R * S * SP;
  if (int rc = c->scratch.ensure(tab + 2 * mat)) return rc;
  double *d_tab = c->scratch.p, *d_l = d_tab + tab, *d_r = d_l + mat;
  HIP_TRY(hipMemsetAsync(d_tab, 0, tab * sizeof(double), c->stream)); // (entries between the rows of a padded index stay 0)
  HIP_TRY(hipMemcpyAsync(d_l, left_host, mat * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_r, right_host, mat * sizeof(double), hipMemcpyHostToDevice, c->stream));
  const unsigned long long *tm = nullptr;
  if (S != 4)
  {
    if (!tipmap_host) return fail(PLLGPU_EINVAL, "a lookup table of %u states needs the code -> state mask map", S);
    if (int rc = c->tipmap.ensure(256)) return rc;
    HIP_TRY(hipMemcpyAsync(c->tipmap.p, tipmap_host, ncodes * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    tm = c->tipmap.p;
  }
  const size_t n = (size_t)ncodes * ncodes * span;
  hipLaunchKernelGGL(k_create_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_tab, d_l, d_r, tm, S, SP, R, ncodes, shift);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(lookup_host, d_tab, tab * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pllgpu_tt_from_lookup(pllgpu_ctx_t *c, double *parent_host, const unsigned char *left_codes, const unsigned char *right_codes,
                                     const double *lookup_host, unsigned sites, unsigned ncodes)
{
  CHECK_CTX(c);
  const unsigned S = c->gg.S, SP = c->gg.SP, R = c->gg.R, span = R * SP, shift = lookup_shift(S, ncodes);
  if (S == 4) ncodes = 16;
  if (!ncodes || ncodes > 256 || !sites) return fail(PLLGPU_EINVAL, "tip-tip update over %u codes, %u sites", ncodes, sites);
  for (unsigned n = 0; n < sites; ++n)
    if (left_codes[n] >= ncodes || right_codes[n] >= ncodes) return fail(PLLGPU_EINVAL, "site %u: tip code beyond the lookup table's %u codes", n, ncodes);
  const size_t entries = (size_t)((ncodes - 1u) << shift) + ncodes, tab = entries * span, out = (size_t)sites * span;
  if (int rc = c->scratch.ensure(tab + out + (2 * (size_t)sites + 7) / 8 + 1)) return rc;
  double *d_tab = c->scratch.p, *d_out = d_tab + tab;
  unsigned char *d_l = reinterpret_cast<unsigned char *>(d_out + out), *d_r = d_l + sites;
  HIP_TRY(hipMemcpyAsync(d_tab, lookup_host, tab * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_l, left_codes, sites, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_r, right_codes, sites, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_tt_from_lookup, dim3((unsigned)((out + 255) / 256)), dim3(256), 0, c->stream, d_out, d_tab, d_l, d_r, sites, span, shift);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(parent_host, d_out, out * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- the exchange of a site-sharded run ---------------------------------------------------------
// a collective library enqueues on the context's stream from the HOST side of the boundary (group.c: ncclAllReduce):
// the calling thread's current device must be the context's while it does (its own may be another, e.g. with
// PLL_AMD_DEVICE=auto); *previous = what pllgpu_leave_device puts back, -1 = nothing to do
extern "C" int pllgpu_enter_device(pllgpu_ctx_t *c, int *previous)
{
  if (!c || !previous) return fail(PLLGPU_EINVAL, "null context");
  *previous = -1;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) cur = -1;
  if (cur == c->device) return 0;
  HIP_TRY(hipSetDevice(c->device));
  *previous = cur;
  return 0;
}

extern "C" void pllgpu_leave_device(int previous)
{
  if (previous >= 0) (void)hipSetDevice(previous);
}

extern "C" double *pllgpu_reduce_buffer(pllgpu_ctx_t *c)
{
  if (!c) return nullptr;
  DeviceScope device_scope_(c);
  if (device_scope_.rc || c->reduce.ensure(2)) return nullptr;
  return c->reduce.p;
}

// a rank whose evaluation failed still brings an operand to the collective (-inf poisons the sum: every rank learns
// of the failure instead of waiting inside the all-reduce)
__global__ void k_set_pair(double *__restrict__ pair, double value, double sequence)
{
  pair[0] = value;
  pair[1] = sequence;
}

extern "C" int pllgpu_reduce_poison(pllgpu_ctx_t *c, double sequence)
{
  CHECK_CTX(c);
  if (!c->reduce.p) return fail(PLLGPU_EINVAL, "no reduce buffer");
  hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, c->stream, c->reduce.p, -HUGE_VAL, sequence);
  HIP_TRY(hipGetLastError());
  return 0;
}

// value, then - once the value has been performed - the sequence word: the order the host's poll relies on
__global__ void k_publish_pair(const double *__restrict__ pair, double *__restrict__ host)
{
  const double v = pair[0], q = pair[1];
  __hip_atomic_store(host, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(host + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int pllgpu_reduce_fetch(pllgpu_ctx_t *c, double expected_sequence, double *value_out, int timeout_ms)
{
  CHECK_CTX(c);
  if (!c->reduce.p) return fail(PLLGPU_EINVAL, "no reduce buffer");
  // words 4 and 5 of the mapped result block (0..2 belong to the synchronous evaluations and the derivatives)
  hipLaunchKernelGGL(k_publish_pair, dim3(1), dim3(1), 0, c->stream, c->reduce.p, c->result_dev + 4);
  HIP_TRY(hipGetLastError());
  unsigned long long want;
  memcpy(&want, &expected_sequence, sizeof want);
  volatile double *res = c->result_host;
  const auto t0 = std::chrono::steady_clock::now();
  const auto arrived = [&] { return __atomic_load_n((const unsigned long long *)&res[5], __ATOMIC_ACQUIRE) == want; };
  unsigned spins = 0;
  bool slow = false;
  while (!arrived())
  {
    if (!slow)
    {
      if ((++spins & 1023u) != 0 || std::chrono::steady_clock::now() - t0 <= std::chrono::milliseconds(50)) continue;
      slow = true; // not the microseconds a collective takes: from here on look at the stream as well, and yield the core
    }
    // A collective a peer never joins does not finish, and hipStreamSynchronize on its stream would never return
    // (ADVICE r3): ask the stream instead, for a bounded time.
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess)
    {
      if (arrived()) break;
      return fail(PLLGPU_ERUNTIME, "the reduced sequence word is %.17g, expected %.17g: the ranks are out of step", (double)res[5], expected_sequence);
    }
    if (q != hipErrorNotReady) HIP_TRY(q);
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 60000))
      return fail(PLLGPU_ERUNTIME, "the all-reduce did not complete within %d ms: a rank of the communicator never joined it (the partition's stream stays blocked behind the collective)",
                  timeout_ms > 0 ? timeout_ms : 60000);
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  *value_out = c->result_host[4];
  return 0;
}

// ---- stream / timing ---------------------------------------------------------------------------
extern "C" int pllgpu_set_stream(pllgpu_ctx_t *c, void *s)
{
  CHECK_CTX(c);
  HIP_TRY(stream_wait(c));
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)s;
  c->own_stream = false;
  return 0;
}

extern "C" void *pllgpu_get_stream(const pllgpu_ctx_t *cc)
{
  // a caller who asks for the stream is about to order its own work against ours: nothing may be held back
  pllgpu_ctx *c = const_cast<pllgpu_ctx *>(cc);
  if (c && (!c->deferred.empty() || c->chain_held))
  {
    DeviceScope device_scope_(c);
    if (device_scope_.rc == 0) (void)flush_deferred(c);
  }
  return c ? (void *)c->stream : nullptr;
}

extern "C" int pllgpu_synchronize(pllgpu_ctx_t *c)
{
  CHECK_CTX(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pllgpu_timer_start(pllgpu_ctx_t *c)
{
  CHECK_CTX(c);
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return 0;
}

extern "C" double pllgpu_timer_stop(pllgpu_ctx_t *c)
{
  if (!c) return -1.0;
  DeviceScope device_scope_(c);
  if (device_scope_.rc) return -1.0;
  if ((!c->deferred.empty() || c->chain_held) && flush_deferred(c)) return -1.0;
  float ms = 0;
  if (hipEventRecord(c->ev1, c->stream) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess ||
      hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess)
  {
    fail(PLLGPU_ERUNTIME, "event timing failed");
    return -1.0;
  }
  return (double)ms;
}

extern "C" unsigned pllgpu_last_launch_count(const pllgpu_ctx_t *c) { return c ? c->last_launches : 0; }
extern "C" unsigned long long pllgpu_plan_replays(const pllgpu_ctx_t *c) { return c ? c->plan_replays : 0ull; }
extern "C" unsigned long long pllgpu_class_map_work(const pllgpu_ctx_t *c, int launches) { return !c ? 0ull : launches ? c->rep_launches_total : c->rep_ops_total; }

// ---- branch-length derivatives ------------------------------------------------------------------
extern "C" int pllgpu_eigenvals_upload(pllgpu_ctx_t *c, unsigned index, const double *host)
{
  CHECK_CTX(c);
  if (index >= c->geo.rate_matrices) return fail(PLLGPU_EINVAL, "eigenvalue set %u out of range", index);
  if (int rc = c->eigenvals.ensure((size_t)c->geo.rate_matrices * c->geo.states_padded)) return rc;
  HIP_TRY(hipMemcpyAsync(c->eigenvals.p + (size_t)index * c->geo.states_padded, host,
                         c->geo.states_padded * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}

extern "C" int pllgpu_rates_upload(pllgpu_ctx_t *c, const double *host)
{
  CHECK_CTX(c);
  if (int rc = c->rates.ensure(c->geo.rate_cats)) return rc;
  HIP_TRY(hipMemcpyAsync(c->rates.p, host, c->geo.rate_cats * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}

static int sumtable_slot(pllgpu_ctx *c, unsigned slot)
{
  if (slot >= PLLGPU_SUMTABLE_SLOTS) return fail(PLLGPU_EINVAL, "sumtable slot %u out of range", slot);
  return c->sumtable[slot].ensure(clv_elems(c, c->geo.sites_alloc));
}

extern "C" int pllgpu_update_sumtable(pllgpu_ctx_t *c, const pllgpu_sumtable_t *st, unsigned slot)
{
  CHECK_CTX(c);
  if (int rc = sumtable_slot(c, slot)) return rc;
  const pllgpu_geometry_t &g = c->geo;
  if (st->left_clv >= g.nodes || st->right_clv >= g.nodes) return fail(PLLGPU_EINVAL, "sumtable references a CLV out of range");
  OpPack pack;
  DevOp &d = pack.ops[0];
  memset(&d, 0, sizeof d);
  d.parent = c->sumtable[slot].p;
  d.entries = g.sites_alloc; // with ascertainment bias the per-state extra entries belong to the table
  if (st->left_is_tip)
  {
    if (st->left_clv >= g.tips || !c->tipchars[st->left_clv].p) return fail(PLLGPU_EINVAL, "tip %u has no codes on the device", st->left_clv);
    d.ltip = c->tipchars[st->left_clv].p;
  }
  else
  {
    if (!c->clv[st->left_clv].p) return fail(PLLGPU_EINVAL, "CLV %u unavailable on the device", st->left_clv);
    d.left = c->clv[st->left_clv].p;
  }
  if (!c->clv[st->right_clv].p) return fail(PLLGPU_EINVAL, "CLV %u unavailable on the device", st->right_clv);
  d.right = c->clv[st->right_clv].p;
  d.lmat = c->pmat.p + (size_t)g.prob_matrices * c->pm_stride;
  d.rmat = c->pmat.p + (size_t)(g.prob_matrices + 1) * c->pm_stride;
  d.layout = ((!st->left_is_tip && c->clv_aos[st->left_clv]) ? kAosLeft : 0u) | (c->clv_aos[st->right_clv] ? kAosRight : 0u);
  if (d.layout && !st->gather) return fail(PLLGPU_EINVAL, "a class-compressed CLV met a sumtable without the gather flag");
  if (st->gather)
  {
    d.lsid = c->ids[st->left_clv] ? wide_map(c, st->left_clv) : nullptr;
    d.rsid = c->ids[st->right_clv] ? wide_map(c, st->right_clv) : nullptr;
  }
  const unsigned kind = st->left_is_tip ? 1u : 0u;
  if (int rc = launch_partials(c, pack, 1, g.sites_alloc, kind, st->gather != 0)) return rc;
  if (g.per_rate_scalers && (st->left_scaler >= 0 || st->right_scaler >= 0))
  {
    DevExcess e;
    memset(&e, 0, sizeof e);
    e.table = c->sumtable[slot].p;
    if (int rc = scaler_ptr(c, st->left_is_tip ? -1 : st->left_scaler, e.pscaler)) return rc;
    if (int rc = scaler_ptr(c, st->right_scaler, e.cscaler)) return rc;
    e.psid = d.lsid;
    e.csid = d.rsid;
    e.sites = g.sites_alloc;
    const unsigned tiles = (g.sites_alloc + 63) / 64;
    hipLaunchKernelGGL(k_sumtable_excess, dim3((tiles + 3) / 4), dim3(256), 0, c->stream, e, c->gg);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int pllgpu_sumtable_upload(pllgpu_ctx_t *c, unsigned slot, const double *host)
{
  CHECK_CTX(c);
  if (int rc = sumtable_slot(c, slot)) return rc;
  const size_t n = (size_t)c->geo.sites_alloc * c->span;
  if (int rc = c->scratch.ensure(n)) return rc;
  HIP_TRY(hipMemcpyAsync(c->scratch.p, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_aos_to_tiled, dim3(1024), dim3(256), 0, c->stream, c->scratch.p, c->sumtable[slot].p, c->geo.sites_alloc,
                     c->gg.S, c->gg.SP, c->gg.R);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int pllgpu_sumtable_download(pllgpu_ctx_t *c, unsigned slot, double *host)
{
  CHECK_CTX(c);
  if (slot >= PLLGPU_SUMTABLE_SLOTS || !c->sumtable[slot].p) return fail(PLLGPU_EINVAL, "sumtable slot %u is empty", slot);
  const size_t n = (size_t)c->geo.sites_alloc * c->span;
  if (int rc = c->scratch.ensure(n)) return rc;
  hipLaunchKernelGGL(k_tiled_to_aos, dim3(1024), dim3(256), 0, c->stream, c->sumtable[slot].p, c->scratch.p, c->geo.sites_alloc,
                     c->gg.S, c->gg.SP, c->gg.R);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(host, c->scratch.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pllgpu_sumtable_release(pllgpu_ctx_t *c, unsigned slot)
{
  CHECK_CTX(c);
  if (slot >= PLLGPU_SUMTABLE_SLOTS) return fail(PLLGPU_EINVAL, "sumtable slot %u out of range", slot);
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->sumtable[slot].release();
  return 0;
}

extern "C" int pllgpu_likelihood_derivatives(pllgpu_ctx_t *c, unsigned slot, double branch_length,
                                             const unsigned *params_indices, unsigned eval_sites, double *d_f, double *dd_f)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  if (slot >= PLLGPU_SUMTABLE_SLOTS || !c->sumtable[slot].p) return fail(PLLGPU_EINVAL, "sumtable slot %u is empty", slot);
  if (eval_sites == 0 || eval_sites > g.sites_alloc) return fail(PLLGPU_EINVAL, "eval_sites %u out of range", eval_sites);
  if (!c->eigenvals.p || !c->rates.p) return fail(PLLGPU_EINVAL, "eigenvalues / category rates were not uploaded");
  if (int rc = c->diag.ensure((size_t)g.rate_cats * g.states * 4)) return rc;
  DevDiag dg;
  DevDeriv dv;
  memset(&dg, 0, sizeof dg);
  memset(&dv, 0, sizeof dv);
  for (unsigned k = 0; k < g.rate_cats; ++k)
  {
    if (params_indices[k] >= g.rate_matrices) return fail(PLLGPU_EINVAL, "params_indices[%u] out of range", k);
    dg.fidx[k] = dv.fidx[k] = (unsigned char)params_indices[k];
  }
  dg.diag = c->diag.p;
  dg.eigenvals = c->eigenvals.p;
  dg.rates = c->rates.p;
  dg.prop_invar = c->prop_invar.p;
  dg.branch_length = branch_length;
  dg.S = g.states;
  dg.SP = g.states_padded;
  dg.R = g.rate_cats;
  // small tables are formed inside k_derivatives (one launch per evaluation); large ones keep the pre-kernel
  const size_t diag_bytes = (size_t)g.rate_cats * g.states * 4 * sizeof(double);
  const unsigned local_diag = diag_bytes <= 32768 ? 1u : 0u;
  if (!local_diag) hipLaunchKernelGGL(k_diagtable, dim3(1), dim3(256), 0, c->stream, dg);

  dv.table = c->sumtable[slot].p;
  dv.diag = c->diag.p;
  dv.freqs = c->freqs.p;
  dv.rate_weights = c->rate_weights.p;
  dv.prop_invar = c->prop_invar.p;
  dv.pattern_weights = c->pattern_weights.p;
  dv.invariant = (c->invariant_set && eval_sites <= g.sites) ? c->invariant.p : nullptr;
  dv.block_sums = c->block_sums.p;
  dv.counter = c->counter.p;
  dv.result = c->result_dev;
  c->seq += 1.0;
  dv.sequence = c->seq;
  dv.sites = eval_sites;
  dv.fenced = c->fenced;
  unsigned long long seq_bits;
  memcpy(&seq_bits, &c->seq, sizeof seq_bits);
  const unsigned tiles = (eval_sites + 63) / 64;
  const unsigned tpw = (tiles + 4 * 1024 - 1) / (4 * 1024);
  const unsigned blocks = (tiles + 4 * tpw - 1) / (4 * tpw);
  hipLaunchKernelGGL(k_derivatives, dim3(blocks), dim3(256), local_diag ? diag_bytes : 0, c->stream, dv, c->gg, tpw, dg, local_diag);
  HIP_TRY(hipGetLastError());
  volatile double *res = c->result_host;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n((const unsigned long long *)&res[1], __ATOMIC_ACQUIRE) != seq_bits)
  {
    if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
    {
      HIP_TRY(hipStreamSynchronize(c->stream));
      break;
    }
  }
  *d_f = c->result_host[0];
  *dd_f = c->result_host[2];
  return 0;
}

// ---- ascertainment-bias terms --------------------------------------------------------------------
extern "C" int pllgpu_asc_terms(pllgpu_ctx_t *c, const pllgpu_edge_t *ed, int is_root, double *terms, unsigned *scalings)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  if (g.sites_alloc < g.sites + g.states) return fail(PLLGPU_EINVAL, "the partition has no per-state extra entries");
  if (ed->parent_clv >= g.nodes || !c->clv[ed->parent_clv].p) return fail(PLLGPU_EINVAL, "CLV %u unavailable on the device", ed->parent_clv);
  DevAsc a;
  memset(&a, 0, sizeof a);
  a.parent = c->clv[ed->parent_clv].p;
  if (int rc = scaler_ptr(c, ed->parent_scaler, a.pscaler)) return rc;
  if (!is_root)
  {
    if (ed->child_clv >= g.nodes || ed->matrix >= g.prob_matrices) return fail(PLLGPU_EINVAL, "edge references an index out of range");
    if (ed->child_is_tip)
    {
      if (ed->child_clv >= g.tips || !c->tipchars[ed->child_clv].p) return fail(PLLGPU_EINVAL, "tip %u has no codes on the device", ed->child_clv);
      a.ctip = c->tipchars[ed->child_clv].p;
    }
    else
    {
      if (!c->clv[ed->child_clv].p) return fail(PLLGPU_EINVAL, "CLV %u unavailable on the device", ed->child_clv);
      a.child = c->clv[ed->child_clv].p;
      if (int rc = scaler_ptr(c, ed->child_scaler, a.cscaler)) return rc;
    }
    a.mat = c->pmat.p + (size_t)ed->matrix * c->pm_stride;
  }
  for (unsigned k = 0; k < g.rate_cats; ++k)
  {
    if (ed->freqs_indices[k] >= g.rate_matrices) return fail(PLLGPU_EINVAL, "freqs_indices[%u] out of range", k);
    a.fidx[k] = (unsigned char)ed->freqs_indices[k];
  }
  a.freqs = c->freqs.p;
  a.rate_weights = c->rate_weights.p;
  a.out = c->result_dev + kAscOff;
  a.first = g.sites;
  a.per_rate = g.per_rate_scalers ? 1 : 0;
  a.is_root = is_root;
  const unsigned long long *tm = c->tipmap_set ? c->tipmap.p : nullptr;
  hipLaunchKernelGGL(k_asc_terms, dim3(g.states), dim3(64), 0, c->stream, a, c->gg, tm);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (unsigned n = 0; n < g.states; ++n)
  {
    terms[n] = c->result_host[kAscOff + n];
    scalings[n] = (unsigned)c->result_host[kAscOff + g.states + n];
  }
  return 0;
}

extern "C" int pllgpu_asc_derivative_terms(pllgpu_ctx_t *c, unsigned slot, int parent_scaler, int child_scaler,
                                           const unsigned *params_indices, double *lk, unsigned *scalings)
{
  CHECK_CTX(c);
  (void)params_indices; // the diag table of the last evaluation already carries them
  const pllgpu_geometry_t &g = c->geo;
  if (g.sites_alloc < g.sites + g.states) return fail(PLLGPU_EINVAL, "the partition has no per-state extra entries");
  if (slot >= PLLGPU_SUMTABLE_SLOTS || !c->sumtable[slot].p || !c->diag.p) return fail(PLLGPU_EINVAL, "no derivative evaluation precedes the ascertainment terms");
  DevAscDeriv a;
  memset(&a, 0, sizeof a);
  a.table = c->sumtable[slot].p;
  a.diag = c->diag.p;
  a.rate_weights = c->rate_weights.p;
  if (int rc = scaler_ptr(c, parent_scaler, a.pscaler)) return rc;
  if (int rc = scaler_ptr(c, child_scaler, a.cscaler)) return rc;
  a.out = c->result_dev + kAscOff;
  a.first = g.sites;
  a.per_rate = g.per_rate_scalers ? 1 : 0;
  hipLaunchKernelGGL(k_asc_deriv_terms, dim3(1), dim3(64), 0, c->stream, a, c->gg);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (unsigned n = 0; n < g.states; ++n)
  {
    lk[3 * n + 0] = c->result_host[kAscOff + 3 * n + 0];
    lk[3 * n + 1] = c->result_host[kAscOff + 3 * n + 1];
    lk[3 * n + 2] = c->result_host[kAscOff + 3 * n + 2];
    scalings[n] = (unsigned)c->result_host[kAscOff + 3 * g.states + n];
  }
  return 0;
}

// ---- transition matrices on the device -----------------------------------------------------------
extern "C" int pllgpu_eigen_upload(pllgpu_ctx_t *c, unsigned index, const double *eigenvecs, const double *inv_eigenvecs,
                                   const double *eigenvals)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  if (index >= g.rate_matrices) return fail(PLLGPU_EINVAL, "eigensystem %u out of range", index);
  const size_t n = (size_t)g.states * g.states_padded;
  if (int rc = c->evecs.ensure(n * g.rate_matrices)) return rc;
  if (int rc = c->ievecs.ensure(n * g.rate_matrices)) return rc;
  HIP_TRY(hipMemcpyAsync(c->evecs.p + index * n, eigenvecs, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->ievecs.p + index * n, inv_eigenvecs, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return pllgpu_eigenvals_upload(c, index, eigenvals);
}

extern "C" int pllgpu_update_pmatrices(pllgpu_ctx_t *c, const unsigned *params_indices, const unsigned *matrix_indices,
                                       const double *branch_lengths, unsigned count)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  if (!count) return 0;
  if (!c->evecs.p || !c->ievecs.p || !c->eigenvals.p || !c->rates.p)
    return fail(PLLGPU_EINVAL, "eigensystem / category rates were not uploaded");
  DevPmat d;
  memset(&d, 0, sizeof d);
  for (unsigned k = 0; k < g.rate_cats; ++k)
  {
    if (params_indices[k] >= g.rate_matrices) return fail(PLLGPU_EINVAL, "params_indices[%u] out of range", k);
    d.fidx[k] = (unsigned char)params_indices[k];
  }
  for (unsigned i = 0; i < count; ++i)
    if (matrix_indices[i] >= g.prob_matrices || !(branch_lengths[i] >= 0))
      return fail(PLLGPU_EINVAL, "matrix index %u / branch length %g invalid", matrix_indices[i], branch_lengths[i]);
  for (unsigned i = 0; i < count; ++i) ++c->pm_version[matrix_indices[i]];
  const bool few = count <= kPmatInline;
  if (!few)
  {
    // (a reused staging buffer is overwritten in stream order, behind the launches that still read it)
    if (int rc = c->mindex.ensure(count)) return rc;
    if (int rc = c->brlen.ensure(count)) return rc;
    HIP_TRY(copy_up(c, c->mindex.p, matrix_indices, count * sizeof(unsigned)));
    HIP_TRY(copy_up(c, c->brlen.p, branch_lengths, count * sizeof(double)));
  }
  d.pmat = c->pmat.p;
  d.evecs = c->evecs.p;
  d.ievecs = c->ievecs.p;
  d.evals = c->eigenvals.p;
  d.rates = c->rates.p;
  d.prop_invar = c->prop_invar.p;
  d.mindex = c->mindex.p;
  d.brlen = c->brlen.p;
  d.pm_stride = c->pm_stride;
  d.S = g.states;
  d.SP = g.states_padded;
  d.SPT = c->gg.SPT;
  const size_t lds = (size_t)2 * g.states * (g.states | 1u) * sizeof(double);
  if (few)
  {
    DevPmatFew f;
    f.d = d;
    for (unsigned i = 0; i < count; ++i)
    {
      f.t[i] = branch_lengths[i];
      f.mi[i] = matrix_indices[i];
    }
    raise_lds_limit((const void *)k_pmatrix_few, c->device, lds);
    hipLaunchKernelGGL(k_pmatrix_few, dim3(count, g.rate_cats), dim3(256), lds, c->stream, f);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  raise_lds_limit((const void *)k_pmatrix, c->device, lds);
  for (unsigned first = 0; first < count; first += 65535u) // gridDim.x stays far below its limit; y = rate
  {
    const unsigned nb = std::min(count - first, 65535u);
    DevPmat dd = d;
    dd.mindex += first;
    dd.brlen += first;
    hipLaunchKernelGGL(k_pmatrix, dim3(nb, g.rate_cats), dim3(256), lds, c->stream, dd);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int pllgpu_pmatrix_download(pllgpu_ctx_t *c, unsigned index, double *host)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  if (index >= g.prob_matrices) return fail(PLLGPU_EINVAL, "matrix %u out of range", index);
  HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier upload may still read the staging vector
  c->stage.resize(c->pm_stride);
  HIP_TRY(hipMemcpyAsync(c->stage.data(), c->pmat.p + (size_t)index * c->pm_stride, c->pm_stride * sizeof(double),
                         hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const unsigned S = g.states, SP = g.states_padded, SPT = c->gg.SPT;
  for (unsigned k = 0; k < g.rate_cats; ++k)
    for (unsigned i = 0; i < S; ++i)
    {
      double *row = host + ((size_t)k * S + i) * SP;
      for (unsigned j = 0; j < S; ++j) row[j] = c->stage[((size_t)k * S + j) * SPT + i];
      for (unsigned j = S; j < SP; ++j) row[j] = 0.0;
    }
  return 0;
}

// ---- site-repeats class maps on the device -------------------------------------------------------
extern "C" int pllgpu_repeats_set_ids(pllgpu_ctx_t *c, unsigned node, unsigned ids)
{
  CHECK_CTX(c);
  if (node >= c->geo.nodes) return fail(PLLGPU_EINVAL, "node %u out of range", node);
  if (c->ids[node] != ids)
  {
    ++c->maps_epoch;
    ++c->maps_version;
    ++c->maps_foreign;
  }
  c->ids[node] = ids;
  if (!ids)
  {
    // (the host may settle a parent that cannot be compressed without a class-map call: what such a call leaves behind
    // for an uncompressed parent is left behind here)
    if (c->rep_left[node] != -1 || c->rep_right[node] != -1 || c->map_forms[node] != 0) ++c->maps_epoch;
    c->rep_left[node] = c->rep_right[node] = -1;
    c->map_forms[node] = 0;
    c->map_widened[node] = 0;
  }
  return 0;
}

extern "C" int pllgpu_repeats_download(pllgpu_ctx_t *c, unsigned node, unsigned *site_id, unsigned *id_site, unsigned ids)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  const unsigned *wide = node < g.nodes ? wide_map(c, node) : nullptr;
  if (!wide || !c->id_site[node].p || ids > g.sites) return fail(PLLGPU_EINVAL, "node %u has no class maps on the device", node);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(site_id, wide, g.sites * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  if (ids) HIP_TRY(hipMemcpyAsync(id_site, c->id_site[node].p, ids * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// Class maps of a whole op list: every dependency level goes through k_rep_mark + k_rep_assign, the levels back to back,
// and the host reads all class counts once at the end (kernels_repeats.h). What the host must know BEFORE a level's
// counts exist - how large a table slice an op may need - comes from upper bounds: an op's classes are at most its
// cells, its cells at most the product of its children's bounds, and a compressed parent's table is smaller than
// lookup_size by the rule itself.
extern "C" int pllgpu_repeats_classes(pllgpu_ctx_t *c, const pllgpu_repop_t *ops, unsigned count, unsigned lookup_size, unsigned *counts_out)
{
  CHECK_CTX(c);
  const pllgpu_geometry_t &g = c->geo;
  const unsigned sites = g.sites;
  if (!count) return 0;
  if (count > c->rep_host_cap) return fail(PLLGPU_EINVAL, "class maps of %u ops in one call (at most %u)", count, c->rep_host_cap);
  const unsigned words = (sites + 31u) / 32u;
  const size_t wstride = ((size_t)words + kRepScanChunk - 1u) / kRepScanChunk * kRepScanChunk;
  // k_rep_bits: 4 ... 32 ranges of the sites per op, each a multiple of 1024 bitmap words (its LDS)
  const unsigned bit_ranges = std::max(4u, std::min(kRepBitsMaxRanges, (words + 2047u) / 2048u));
  const unsigned bit_words = ((words + bit_ranges - 1u) / bit_ranges + kRepScanThreads - 1u) / kRepScanThreads * kRepScanThreads;
  const size_t op_scratch = 2u * wstride + 2u * kRepBitsMaxRanges; // per op of a launch: bitmap, running counts, the ranges' totals and starts
  const size_t table_cap = (size_t)64 << 20; // cells per launch (256 MB); a single larger op still gets its slice
  const size_t melems = map_elems(c);
  // How many of the levels to launch. The kernels decide by themselves which parents are compressed, but every level costs
  // its launches even when none is; where compression ended the last time an op list of this length came is remembered,
  // only the levels up to there are launched, and the rule (src/repeats.c:100-110) is evaluated HERE for the ops above,
  // from the counts that came back: if it admits one of them after all, the whole call is repeated with every level.
  unsigned ncut = count;
  unsigned long long key = 1469598103934665603ull; // FNV-1a over what identifies the list (two lists of one length differ in where compression ends)
  for (unsigned i = 0; i < count; ++i)
    for (unsigned w : {ops[i].parent, ops[i].left, ops[i].right, ops[i].level})
      key = (key ^ w) * 1099511628211ull;
  if (c->rep_hints && c->rep_hint_count == count && c->rep_hint_key == key)
  {
    ncut = 0;
    while (ncut < count && ops[ncut].level <= c->rep_hint_level && c->rep_hint_any) ++ncut;
  }

  // bounds, checks, buffers - everything that may allocate (and so synchronise) before the first launch
  std::vector<unsigned long long> ub_cells(count);
  std::vector<unsigned> ub_classes(count);
  for (unsigned i = 0; i < count; ++i)
  {
    const pllgpu_repop_t &o = ops[i];
    if (o.parent >= g.nodes || o.left >= g.nodes || o.right >= g.nodes) return fail(PLLGPU_EINVAL, "repeats op references a node out of range");
    if (i && o.level < ops[i - 1].level) return fail(PLLGPU_EINVAL, "repeats ops are not sorted by level");
    if ((o.lsrc >= 0 && ((unsigned)o.lsrc >= i || ops[o.lsrc].level >= o.level || ops[o.lsrc].parent != o.left)) ||
        (o.rsrc >= 0 && ((unsigned)o.rsrc >= i || ops[o.rsrc].level >= o.level || ops[o.rsrc].parent != o.right)))
      return fail(PLLGPU_EINVAL, "repeats op %u: a child's producer is not an op of a lower level", i);
    const unsigned long long bl = o.lsrc >= 0 ? ub_classes[o.lsrc] : o.nleft, br = o.rsrc >= 0 ? ub_classes[o.rsrc] : o.nright;
    unsigned long long cells = bl * br;
    if (o.force)
    {
      if (o.lsrc >= 0 || o.rsrc >= 0) return fail(PLLGPU_EINVAL, "repeats op %u: a decision taken by the host needs both children's counts", i);
      if (!cells) return fail(PLLGPU_EINVAL, "repeats op: children %u / %u have no class maps on the device", o.left, o.right);
    }
    else if (cells >= lookup_size)
      cells = lookup_size ? lookup_size - 1u : 0u;
    if (cells >= 0x7FFFFFFFull) return fail(PLLGPU_EINVAL, "repeats table of %llu cells exceeds 31-bit addressing", cells);
    ub_cells[i] = cells;
    ub_classes[i] = (unsigned)std::min<unsigned long long>(cells, sites);
    // children that come from outside the call: their maps in the form their class count asks for
    const unsigned kid[2] = {o.left, o.right}, kn[2] = {o.nleft, o.nright};
    const int ksrc[2] = {o.lsrc, o.rsrc};
    for (int k = 0; k < 2 && cells && i < ncut; ++k)
    {
      if (ksrc[k] >= 0 || !kn[k]) continue;
      if (kn[k] <= kRepNarrow)
      {
        if (int rc = narrow_map(c, kid[k])) return rc;
      }
      else if (!wide_map(c, kid[k]))
        return fail(PLLGPU_EINVAL, "repeats op: child %u has no class maps on the device", kid[k]);
    }
    if (!cells || i >= ncut) continue; // cannot be compressed / not launched: nothing to write
    if (int rc = c->site_id8[o.parent].ensure(melems)) return rc;
    if (ub_classes[i] > kRepNarrow)
      if (int rc = c->site_id[o.parent].ensure(melems)) return rc;
    if (int rc = c->id_site[o.parent].ensure(ub_classes[i])) return rc;
    if (int rc = c->lent[o.parent].ensure(ub_classes[i])) return rc;
    if (int rc = c->rent[o.parent].ensure(ub_classes[i])) return rc;
  }
  HIP_TRY(hipGetLastError());

  // launches: the ops of a level, up to kRepOps and table_cap cells at a time
  struct Launch
  {
    unsigned first, n, wgs, mark_lds, assign_lds;
    bool rank; // some op's table may be a large one: k_rep_fold + k_rep_scan + k_rep_rank between k_rep_mark and k_rep_assign
    bool narrow, general; // the builds of k_rep_mark its ops may need (kernels_repeats.h)
    size_t max_cells;     // the largest table an op of the launch may have
    bool fused;           // some op has fused children
    bool assign;          // some op's site -> class pass is k_rep_assign's (not every one deferred to the parent's k_rep_mark)
  };
  std::vector<Launch> launches;
  std::vector<RepOp> &rops = c->rep_ops_host;
  rops.assign(ncut, RepOp());
  // A child's site -> class pass inside its parent's k_rep_mark (kernels_repeats.h: kRepFuseLeft): for a child X of the call
  // whose table is small for sure, over byte maps, read by exactly one launched op P whose children's maps are both bytes
  std::vector<unsigned> fuse(ncut, 0u);
  std::vector<int> final_slot(ncut, -1);
  unsigned ndeferred = 0;
  if (c->rep_fuse)
  {
    std::vector<unsigned> readers(ncut, 0u);
    for (unsigned i = 0; i < ncut; ++i)
    {
      if (ops[i].lsrc >= 0) ++readers[ops[i].lsrc];
      if (ops[i].rsrc >= 0) ++readers[ops[i].rsrc];
    }
    auto bytes = [&](int src, unsigned given) { return (src >= 0 ? ub_classes[src] : given) <= kRepNarrow; };
    for (unsigned i = 0; i < ncut; ++i)
    {
      const pllgpu_repop_t &o = ops[i];
      if (o.force || !bytes(o.lsrc, o.nleft) || !bytes(o.rsrc, o.nright) || (o.lsrc >= 0 && o.lsrc == o.rsrc)) continue;
      const int kid[2] = {o.lsrc, o.rsrc};
      for (int k = 0; k < 2; ++k)
      {
        const int x = kid[k];
        if (x < 0 || readers[x] != 1u || ops[x].force || !ub_cells[x] || ub_cells[x] > kRepFuseCells) continue;
        if (!bytes(ops[x].lsrc, ops[x].nleft) || !bytes(ops[x].rsrc, ops[x].nright)) continue;
        fuse[i] |= k ? kRepFuseRight : kRepFuseLeft;
        fuse[x] |= kRepDeferred;
        final_slot[x] = (int)ndeferred++;
      }
    }
    if (int rc = c->rep_final.ensure((size_t)std::max(1u, ndeferred) * kRepFuseCells)) return rc;
  }
  size_t arena = 0;
  for (unsigned done = 0; done < ncut;)
  {
    unsigned room = 0;
    while (done + room < ncut && room < (unsigned)kRepOps && ops[done + room].level == ops[done].level) ++room;
    // workgroups per op: ~512 per launch - all resident at once, and on a 125k-site shard measurably better than 1024
    // (how they split into table parts and site ranges: k_rep_mark)
    const unsigned wgs = c->rep_wgs ? c->rep_wgs : std::max(1u, std::min(64u, (512u + room - 1u) / room));
    Launch L = {done, 0, wgs, kRepSmallCells, 64, false, false, false, 0, false, false};
    size_t cells = 0;
    for (unsigned k = 0; k < room; ++k)
    {
      const unsigned i = done + k;
      const size_t ub = (size_t)ub_cells[i];
      // all copies of the table (k_rep_mark): a small one has a copy per workgroup, a large one up to max_ranges - and only
      // while its parts are fewer than the workgroups
      const size_t slice = ((ub <= kRepSmallCells ? ub * wgs : std::max(ub, std::min(ub * c->rep_max_ranges, (size_t)kRepLdsCells * wgs))) + 3u) & ~(size_t)3u;
      if (ub > kRepSmallCells) L.rank = true;
      L.max_cells = std::max(L.max_cells, ub);
      {
        const unsigned long long bl = ops[i].lsrc >= 0 ? ub_classes[ops[i].lsrc] : ops[i].nleft, br = ops[i].rsrc >= 0 ? ub_classes[ops[i].rsrc] : ops[i].nright;
        // the narrow build for the ops that are its for sure; an op that only may turn out so goes with the general build
        if (bl <= kRepNarrow && br <= kRepNarrow && ub <= kRepSmallCells) L.narrow = true;
        else L.general = true;
      }
      if (k && cells + slice > table_cap) break;
      RepOp &r = rops[i];
      const pllgpu_repop_t &o = ops[i];
      r.l8 = c->site_id8[o.left].p;
      r.r8 = c->site_id8[o.right].p;
      r.l32 = c->site_id[o.left].p;
      r.r32 = c->site_id[o.right].p;
      r.p8 = c->site_id8[o.parent].p;
      r.p32 = c->site_id[o.parent].p;
      r.pids = c->id_site[o.parent].p;
      r.lent = c->lent[o.parent].p;
      r.rent = c->rent[o.parent].p;
      r.table = reinterpret_cast<unsigned *>(cells); // offset for now: the arena may still grow
      r.bitmap = reinterpret_cast<unsigned *>((size_t)k * op_scratch);
      r.keep = nullptr;
      if (ub > kRepSmallCells)
      {
        // (zeroed when it is new: its last word says whether the rest is what the node's maps derive from)
        DevBuf<unsigned> &kb = c->rep_keep[o.parent];
        const size_t had = kb.cap;
        if (int rc = kb.ensure((size_t)words + 1u)) return rc;
        if (kb.cap != had) HIP_TRY(hipMemsetAsync(kb.p, 0, kb.cap * sizeof(unsigned), c->stream));
        r.keep = kb.p;
      }
      r.lsrc = o.lsrc;
      r.rsrc = o.rsrc;
      r.nleft = o.nleft;
      r.nright = o.nright;
      r.slot = i;
      r.force = o.force ? 1u : 0u;
      r.slice = (unsigned)std::min<size_t>(slice, 0x7FFFFFFFu);
      r.flags = fuse[i];
      if (fuse[i] & (kRepFuseLeft | kRepFuseRight)) L.fused = true;
      if (!(fuse[i] & kRepDeferred) && ub) L.assign = true;
      cells += slice;
      L.mark_lds = std::max<unsigned>(L.mark_lds, (unsigned)std::min<size_t>(ub, kRepLdsCells));
      if (ub <= kRepAssignLds) L.assign_lds = std::max<unsigned>(L.assign_lds, (unsigned)((ub + 3u) & ~(size_t)3u));
      ++L.n;
    }
    arena = std::max(arena, cells);
    launches.push_back(L);
    done += L.n;
  }
  if (ncut)
  {
  c->rep_ops_total += ncut;
  if (int rc = c->rep_table.ensure(arena)) return rc;
  {
    // per op of a launch: the bitmap (zero between launches: k_rep_assign clears what k_rep_mark set), then the running counts
    const size_t had = c->rep_blocksum.cap;
    if (int rc = c->rep_blocksum.ensure((size_t)kRepOps * op_scratch)) return rc;
    if (c->rep_blocksum.cap != had || c->rep_scratch_dirty) HIP_TRY(hipMemsetAsync(c->rep_blocksum.p, 0, c->rep_blocksum.cap * sizeof(unsigned), c->stream));
    if (c->rep_scratch_dirty) HIP_TRY(hipMemsetAsync(c->rep_sync.p, 0, c->rep_sync.cap * sizeof(unsigned), c->stream));
    c->rep_scratch_dirty = true; // until this call has gone through
  }
  if (int rc = c->rep_counts.ensure(count)) return rc;
  if (int rc = c->rep_ops.ensure((size_t)count * sizeof(RepOp))) return rc;
  for (unsigned i = 0; i < ncut; ++i)
  {
    rops[i].table = c->rep_table.p + reinterpret_cast<size_t>(rops[i].table);
    rops[i].bitmap = c->rep_blocksum.p + reinterpret_cast<size_t>(rops[i].bitmap);
    rops[i].final = final_slot[i] >= 0 ? c->rep_final.p + (size_t)final_slot[i] * kRepFuseCells : rops[i].table;
  }
  // the descriptors go up unless the device array holds exactly these (the same list over the same buffers: the re-evaluation
  // of one tree - a copy of 14 KB is 5 us on the stream, ahead of the first launch)
  if (c->rep_ops_sent.size() != (size_t)ncut * sizeof(RepOp) || c->rep_ops_sent_at != c->rep_ops.p ||
      memcmp(c->rep_ops_sent.data(), rops.data(), c->rep_ops_sent.size()) != 0)
  {
    // (staged in the context's pinned block; ordered behind the previous call's kernels)
    HIP_TRY(copy_up(c, c->rep_ops.p, rops.data(), (size_t)ncut * sizeof(RepOp)));
    c->rep_ops_sent.assign(reinterpret_cast<const unsigned char *>(rops.data()), reinterpret_cast<const unsigned char *>(rops.data() + ncut));
    c->rep_ops_sent_at = c->rep_ops.p;
  }
  c->rep_host[c->rep_host_cap + 1u] = 0u;
  RepPack pk;
  memset(&pk, 0, sizeof pk);
  pk.counts = c->rep_counts.p;
  pk.tickets = c->rep_sync.p;
  pk.launch_ticket = c->rep_sync.p + kRepOps;
  pk.changed = c->rep_changed.p;
  pk.all_ops = reinterpret_cast<const RepOp *>(c->rep_ops.p);
  pk.max_ranges = c->rep_max_ranges;
  pk.host_counts = c->rep_host_dev;
  pk.ncounts = ncut;
  pk.host_cap = c->rep_host_cap;
  pk.sites = sites;
  pk.lookup = lookup_size;
  pk.wstride = (unsigned)wstride;
  pk.sequence = ++c->rep_seq;
  pk.fenced = c->fenced;
  for (size_t li = 0; li < launches.size(); ++li)
  {
    const Launch &L = launches[li];
    pk.ops = reinterpret_cast<const RepOp *>(c->rep_ops.p) + L.first;
    pk.nops = L.n;
    pk.mark_wgs = L.wgs;
    pk.mark_lds_cells = L.mark_lds;
    const bool last = li + 1 == launches.size();
    pk.has_rank = L.rank ? 1u : 0u;
    pk.publish = last && !L.rank ? 1u : 0u;
    const unsigned n8 = (L.n + 7u) / 8u * 8u;
    pk.has_narrow = L.narrow ? 1u : 0u;
    pk.has_general = L.general || !L.narrow ? 1u : 0u;
    c->rep_launches_total += (pk.has_narrow ? 1u : 0u) + (pk.has_general ? 1u : 0u) + (L.rank ? 3u : 0u) + (L.assign ? 1u : 0u);
    if (pk.has_narrow && L.fused)
      hipLaunchKernelGGL(k_rep_mark_narrow_fused, dim3(n8 * L.wgs), dim3(kRepThreads), kRepSmallCells * sizeof(unsigned), c->stream, pk);
    else if (pk.has_narrow)
      hipLaunchKernelGGL(k_rep_mark_narrow, dim3(n8 * L.wgs), dim3(kRepThreads), kRepSmallCells * sizeof(unsigned), c->stream, pk);
    if (pk.has_general)
    {
      const size_t mark_bytes = (size_t)L.mark_lds * sizeof(unsigned);
      raise_lds_limit((const void *)k_rep_mark, c->device, mark_bytes);
      hipLaunchKernelGGL(k_rep_mark, dim3(n8 * L.wgs), dim3(kRepThreads), mark_bytes, c->stream, pk);
    }
    if (L.rank)
    {
      // the bitmap of first sites: in LDS per (op, range of sites) where the tables are small enough for every range's
      // workgroup to read all cells (k_rep_bits), else by atomics in k_rep_fold and one workgroup per op (k_rep_scan)
      const bool bits = c->rep_bits && L.max_cells <= kRepBitsCells;
      pk.bit_ranges = bits ? bit_ranges : 0u;
      pk.bit_words = bits ? bit_words : 0u;
      hipLaunchKernelGGL(k_rep_fold, dim3(n8 * kRepFoldTiles), dim3(kRepFoldThreads), 0, c->stream, pk);
      pk.publish = last ? 1u : 0u;
      if (bits)
        hipLaunchKernelGGL(k_rep_bits, dim3(n8 * bit_ranges), dim3(kRepScanThreads), (size_t)bit_words * sizeof(unsigned), c->stream, pk);
      else
        hipLaunchKernelGGL(k_rep_scan, dim3(L.n), dim3(kRepScanThreads), 0, c->stream, pk);
      hipLaunchKernelGGL(k_rep_rank, dim3(n8 * kRepRankTiles), dim3(kRepRankThreads), 0, c->stream, pk);
    }
    // k_rep_assign: a workgroup takes 1..4 rounds of 16384 sites - ~512 workgroups per launch, more rounds where a large
    // table has to be brought into LDS first
    pk.lds_cells = L.assign_lds;
    const size_t assign_bytes = (size_t)L.assign_lds * sizeof(unsigned short);
    const unsigned per_round = kRepAssignThreads * 16u, rounds = (sites + per_round - 1u) / per_round;
    pk.assign_iters = std::max(1u, std::min(4u, rounds * L.n / (assign_bytes > 32768 ? 256u : 512u)));
    pk.wgs = (rounds + pk.assign_iters - 1u) / pk.assign_iters;
    raise_lds_limit((const void *)k_rep_assign, c->device, assign_bytes);
    if (L.assign) hipLaunchKernelGGL(k_rep_assign, dim3(n8 * pk.wgs), dim3(kRepAssignThreads), assign_bytes, c->stream, pk);
  }
  HIP_TRY(hipGetLastError());
  // the class counts arrive in mapped host memory as soon as the last k_rep_mark knows them (its k_rep_assign still
  // runs; whatever uses the maps is ordered behind it by the stream): poll the sequence word for a bounded time
  {
    volatile unsigned *seq = c->rep_host + c->rep_host_cap;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != pk.sequence)
      if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50))
      {
        HIP_TRY(hipStreamSynchronize(c->stream));
        break;
      }
    if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != pk.sequence) return fail(PLLGPU_ERUNTIME, "class counts did not arrive");
    if (c->rep_host[c->rep_host_cap + 1u]) return fail(PLLGPU_ERUNTIME, "class maps: a table outgrew what the host had planned for it (%u)", c->rep_host[c->rep_host_cap + 1u]);
  }
  c->rep_scratch_dirty = false;
  }
  // the ops above the launched levels: the rule, from the counts below them
  for (unsigned i = 0; i < ncut; ++i) counts_out[i] = c->rep_host[i];
  for (unsigned i = ncut; i < count; ++i)
  {
    const pllgpu_repop_t &o = ops[i];
    auto ids_of = [&](int src, unsigned given) -> unsigned long long {
      if (src < 0) return given;
      const unsigned w = counts_out[src], n = w & ~kRepFlag;
      return (w & kRepFlag) && n < sites ? n : 0u;
    };
    const unsigned long long nl = ids_of(o.lsrc, o.nleft), nr = ids_of(o.rsrc, o.nright), cells = nl * nr;
    if (o.force || (cells && cells < lookup_size && nl <= sites / 2u && nr <= sites / 2u))
    {
      c->rep_hint_count = 0; // the forecast was wrong: every level this time
      return pllgpu_repeats_classes(c, ops, count, lookup_size, counts_out);
    }
    counts_out[i] = 0u;
  }
  c->rep_hint_count = count;
  c->rep_hint_key = key;
  c->rep_hint_any = false;
  c->rep_hint_level = 0;
  for (unsigned i = 0; i < count; ++i)
    if (counts_out[i] & kRepFlag)
    {
      c->rep_hint_any = true;
      c->rep_hint_level = std::max(c->rep_hint_level, ops[i].level);
    }
  // cached launches (LevelPlan) carry class counts and pointers, not contents: they stay valid when this call found what the
  // last one found - the re-evaluation of one tree, again and again, with the reference's pll_update_partials
  bool reshaped = false;
  for (unsigned i = 0; i < count; ++i)
  {
    const unsigned word = counts_out[i], classes = word & ~kRepFlag;
    const unsigned parent = ops[i].parent;
    if (!(word & kRepFlag))
    {
      reshaped = reshaped || c->rep_left[parent] != -1 || c->rep_right[parent] != -1 || c->map_forms[parent] != 0;
      c->rep_left[parent] = c->rep_right[parent] = -1;
      c->map_forms[parent] = 0;
      c->map_widened[parent] = 0;
      continue;
    }
    if ((unsigned long long)classes > ub_cells[i]) return fail(PLLGPU_ERUNTIME, "class maps: op %u reports %u classes of at most %llu", i, classes, ub_cells[i]);
    const unsigned char form = classes <= kRepNarrow ? kMap8 : kMap32;
    reshaped = reshaped || c->rep_left[parent] != (int)ops[i].left || c->rep_right[parent] != (int)ops[i].right || !(c->map_forms[parent] & form);
    c->rep_left[parent] = (int)ops[i].left;
    c->rep_right[parent] = (int)ops[i].right;
    c->map_forms[parent] = form;
    if (form == kMap8 && c->map_widened[parent])
    {
      // a cached launch reads this node's 32-bit form: bring it up to date behind the bytes
      c->map_widened[parent] = 0;
      if (!wide_map(c, parent)) return fail(PLLGPU_ENOMEM, "class maps: no room for the 32-bit form of node %u", parent);
    }
  }
  if (reshaped) ++c->maps_epoch;
  ++c->maps_version;
  HIP_TRY(hipGetLastError());
  return 0;
}
