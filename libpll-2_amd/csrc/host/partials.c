/* partials.c - pll_update_partials / pll_update_partials_rep: the traversal driver.
 *
 * The reference walks the operation list strictly in order and calls one CPU kernel per op
 * (src/partials.c:245-291). Here the list is (1) classified per op exactly like the reference's
 * case_* functions (:24-235: site repeats / tip-tip / tip-inner / inner-inner, tip moved to the
 * left), (2) cut into dependency LEVELS - ops of one level neither read nor write each other's
 * CLVs or scalers - and (3) handed to the device layer, which launches one kernel per level and
 * child-kind with the level's ops in grid.y. A balanced 64-taxon traversal becomes 5 launches
 * instead of 62. The call is asynchronous; results stay in HBM.
 */
#include "pll_internal.h"

static int grow_scratch(pll_partition_t *p, pll_amd_ext_t *x, unsigned int count)
{
  if (!x->lvl_clv_w)
  {
    x->lvl_clv_w = (int *)malloc(sizeof(int) * (p->nodes + 1));
    x->lvl_clv_r = (int *)malloc(sizeof(int) * (p->nodes + 1));
    x->lvl_sc_w = (int *)malloc(sizeof(int) * (p->scale_buffers + 1));
    x->lvl_sc_r = (int *)malloc(sizeof(int) * (p->scale_buffers + 1));
    if (!x->lvl_clv_w || !x->lvl_clv_r || !x->lvl_sc_w || !x->lvl_sc_r) return 0;
  }
  if (count > x->gops_cap)
  {
    free(x->gops);
    x->gops = (pllgpu_op_t *)malloc(sizeof(pllgpu_op_t) * 2 * count);
    x->gops_cap = x->gops ? count : 0;
    if (!x->gops) return 0;
  }
  return 1;
}

static int imax(int a, int b) { return a > b ? a : b; }

/* Level of each op: one more than the latest earlier op it conflicts with. Conflicts are taken
 * on CLV indices AND scaler indices, read-after-write, write-after-write and write-after-read:
 * the three pll_unode_t records of an unrooted inner node share one clv_index, so partial
 * traversals legally re-orient (overwrite) a CLV that an earlier op of the same list still reads
 * (SURVEY.md section 3.4). */
static unsigned int assign_levels(const pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *ops,
                                  unsigned int count, unsigned int *level, int *war)
{
  unsigned int i, nlevels = 0;
  for (i = 0; i < p->nodes; ++i) x->lvl_clv_w[i] = x->lvl_clv_r[i] = -1;
  for (i = 0; i < p->scale_buffers; ++i) x->lvl_sc_w[i] = x->lvl_sc_r[i] = -1;
  for (i = 0; i < count; ++i)
  {
    const pll_operation_t *o = &ops[i];
    int l = -1;
    /* RAW on inputs */
    l = imax(l, x->lvl_clv_w[o->child1_clv_index]);
    l = imax(l, x->lvl_clv_w[o->child2_clv_index]);
    if (o->child1_scaler_index >= 0) l = imax(l, x->lvl_sc_w[o->child1_scaler_index]);
    if (o->child2_scaler_index >= 0) l = imax(l, x->lvl_sc_w[o->child2_scaler_index]);
    /* WAW + WAR on outputs; kept separately too: the device layer may run an op together with the
     * producers of its children, one level early, if nothing of this kind sits there */
    int w = imax(x->lvl_clv_w[o->parent_clv_index], x->lvl_clv_r[o->parent_clv_index]);
    if (o->parent_scaler_index >= 0)
    {
      w = imax(w, x->lvl_sc_w[o->parent_scaler_index]);
      w = imax(w, x->lvl_sc_r[o->parent_scaler_index]);
    }
    war[i] = w;
    l = imax(l, w);
    l += 1;
    level[i] = (unsigned int)l;
    if ((unsigned int)l + 1 > nlevels) nlevels = l + 1;
    x->lvl_clv_w[o->parent_clv_index] = l;
    x->lvl_clv_r[o->child1_clv_index] = imax(x->lvl_clv_r[o->child1_clv_index], l);
    x->lvl_clv_r[o->child2_clv_index] = imax(x->lvl_clv_r[o->child2_clv_index], l);
    if (o->parent_scaler_index >= 0) x->lvl_sc_w[o->parent_scaler_index] = l;
    if (o->child1_scaler_index >= 0) x->lvl_sc_r[o->child1_scaler_index] = imax(x->lvl_sc_r[o->child1_scaler_index], l);
    if (o->child2_scaler_index >= 0) x->lvl_sc_r[o->child2_scaler_index] = imax(x->lvl_sc_r[o->child2_scaler_index], l);
  }
  return nlevels;
}

/* Site repeats: the class maps of a whole list are computed before its first CLV kernel runs, and the list is classified
 * with the class counts the maps END with. That is the reference's op-by-op order (src/partials.c:253-257: maps, then
 * the CLV, per op) only while no op overwrites a node an earlier op of the list has read or written - true of every
 * traversal for ONE evaluation, not of two traversals handed over as one list (the three records of an inner node share
 * a clv_index: the second may turn round a CLV the first has read - found by tests/test_gpu_tree_search.py). Returns the
 * index of the first op that does so, 0 if none: the caller runs the list in two pieces, one after the other. */
static unsigned int first_overwrite(const pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *ops, unsigned int count)
{
  unsigned int i;
  for (i = 0; i < p->nodes; ++i) x->lvl_clv_w[i] = 0;
  for (i = 0; i < p->scale_buffers; ++i) x->lvl_sc_w[i] = 0;
  for (i = 0; i < count; ++i)
  {
    const pll_operation_t *o = &ops[i];
    if (x->lvl_clv_w[o->parent_clv_index] || (o->parent_scaler_index >= 0 && x->lvl_sc_w[o->parent_scaler_index])) return i;
    x->lvl_clv_w[o->parent_clv_index] = x->lvl_clv_w[o->child1_clv_index] = x->lvl_clv_w[o->child2_clv_index] = 1;
    if (o->parent_scaler_index >= 0) x->lvl_sc_w[o->parent_scaler_index] = 1;
    if (o->child1_scaler_index >= 0) x->lvl_sc_w[o->child1_scaler_index] = 1;
    if (o->child2_scaler_index >= 0) x->lvl_sc_w[o->child2_scaler_index] = 1;
  }
  return 0;
}

static void fail_loudly(const char *what)
{
  fprintf(stderr, "libpll_amd: %s: [%d] %s\n", what, pll_errno, pll_errmsg);
}
#define BAIL()                                \
  do                                          \
  {                                           \
    fail_loudly("pll_update_partials");       \
    return;                                   \
  } while (0)

/* Re-evaluations of one tree - the common case between two topology changes, and bench.py's timed loop - hand over the
 * same operation list again and again. When nothing the device would have to be told about is dirty and nothing the
 * classification depends on has changed since that list last went through the whole function (fast_valid is cleared
 * wherever tip forms, class counts or caller-written arrays change), the classified, level-sorted list of that call
 * (x->gops) is still right: skip levels, flushes and classification (126 ops: ~4 us of host time between one step's
 * result and the next step's first kernel). */
static int nothing_dirty(const pll_partition_t *p, const pll_amd_ext_t *x)
{
  unsigned int i;
  if (x->rate_weights_dirty | x->pattern_weights_dirty | x->invariant_dirty | x->prop_invar_dirty | x->tipmap_dirty) return 0;
  if (memchr(x->freqs_dirty, 1, p->rate_matrices)) return 0;
  for (i = x->fast_lo; i <= x->fast_hi; ++i)
    if (x->pmatrix_dirty[i] && !x->pmatrix_stale[i]) return 0;
  if (memchr(x->tipchars_dirty, 1, p->tips)) return 0;
  if (pll_repeats_enabled(p) && memchr(x->repeats_dirty, 1, p->nodes)) return 0;
  /* a tip the device reads as one-byte codes keeps its indicator CLV in the host mirror for good (tips.c): what the
   * device needs of it is covered by tipchars_dirty above */
  for (i = 0; i < p->tips; ++i)
    if (x->clv_side[i] == SIDE_HOST && !pll_tip_by_codes(p, i)) return 0;
  if (p->nodes > p->tips && memchr(x->clv_side + p->tips, SIDE_HOST, p->nodes - p->tips)) return 0;
  if (p->scale_buffers && memchr(x->scaler_side, SIDE_HOST, p->scale_buffers)) return 0;
  return 1;
}

static void mark_results(pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *ops, unsigned int count)
{
  unsigned int i;
  for (i = 0; i < count; ++i)
  {
    x->clv_side[ops[i].parent_clv_index] = SIDE_DEVICE;
    if (ops[i].parent_scaler_index >= 0)
    {
      x->scaler_side[ops[i].parent_scaler_index] = SIDE_DEVICE;
      x->scaler_entries[ops[i].parent_scaler_index] = pll_get_sites_number(p, ops[i].parent_clv_index);
    }
  }
}

void pll_update_partials(pll_partition_t *p, const pll_operation_t *ops, unsigned int count)
{
  pll_update_partials_rep(p, ops, count, 1);
}

void pll_update_partials_rep(pll_partition_t *p, const pll_operation_t *ops, unsigned int count,
                             unsigned int update_repeats)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  unsigned int i;
  if (x) x->fast_taken = 0; /* pll_gpu_last_update_replayed() speaks of THIS call, also when it returns early */
  if (!count) return;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_update_partials: no MI355X context behind this partition; this library has no CPU path");
    fail_loudly("pll_update_partials");
    return;
  }
  for (i = 0; i < count; ++i)
    if (ops[i].parent_clv_index >= p->nodes || ops[i].child1_clv_index >= p->nodes ||
        ops[i].child2_clv_index >= p->nodes || ops[i].child1_matrix_index >= p->prob_matrices ||
        ops[i].child2_matrix_index >= p->prob_matrices || ops[i].parent_scaler_index >= (int)p->scale_buffers ||
        ops[i].child1_scaler_index >= (int)p->scale_buffers || ops[i].child2_scaler_index >= (int)p->scale_buffers)
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_update_partials: operation %u has an index out of range", i);
      fail_loudly("pll_update_partials");
      return;
    }
  const int rep = pll_repeats_enabled(p);
  unsigned int *level = NULL;
  int *war = NULL;
  unsigned int nlevels = 0;
  if (rep && update_repeats)
  {
    /* site-repeats class maps first: a parent's classes derive from its children's (src/partials.c:256-257) - on the
     * device, every dependency level in one call. When they come out as they were (the same tree evaluated again with
     * the reference's pll_update_partials), the list classified by the last whole pass is still right: fast path below */
    if (!grow_scratch(p, x, count))
    {
      pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_update_partials: out of memory");
      fail_loudly("pll_update_partials");
      return;
    }
    const unsigned int cut = count > 1 ? first_overwrite(p, x, ops, count) : 0;
    if (cut)
    {
      pll_update_partials_rep(p, ops, cut, update_repeats);
      pll_update_partials_rep(p, ops + cut, count - cut, update_repeats);
      return;
    }
    level = (unsigned int *)(x->gops + count); /* second half of the scratch block */
    war = (int *)(level + count);
    nlevels = assign_levels(p, x, ops, count, level, war);
    if (!pll_update_repeats_device(p, x, ops, count, level, nlevels)) BAIL();
  }
  if (x->fast_valid && count == x->fast_count && !x->always_upload && !x->eager_mirror &&
      memcmp(ops, x->fast_ops, (size_t)count * sizeof *ops) == 0 && nothing_dirty(p, x))
  {
    if (pllgpu_update_partials(x->ctx, x->gops, count) != 0)
    {
      x->fast_valid = 0;
      pll_set_gpu_error("pll_update_partials");
      return;
    }
    mark_results(p, x, ops, count);
    x->fast_taken = 1;
    return;
  }
  x->fast_valid = 0;
  x->fast_taken = 0;
  if (!grow_scratch(p, x, count))
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_update_partials: out of memory");
    fail_loudly("pll_update_partials");
    return;
  }

  if (!level)
  {
    level = (unsigned int *)(x->gops + count); /* second half of the scratch block */
    war = (int *)(level + count);
    nlevels = assign_levels(p, x, ops, count, level, war);
  }

  /* 2. bring inputs up to date on the device */
  if (!pll_flush_model(p, x)) BAIL();
  unsigned int lo = p->prob_matrices, hi = 0;
  for (i = 0; i < count; ++i)
  {
    const pll_operation_t *o = &ops[i];
    if (o->child1_matrix_index < lo) lo = o->child1_matrix_index;
    if (o->child2_matrix_index < lo) lo = o->child2_matrix_index;
    if (o->child1_matrix_index > hi) hi = o->child1_matrix_index;
    if (o->child2_matrix_index > hi) hi = o->child2_matrix_index;
  }
  if (!pll_flush_pmatrix(p, x, lo, hi)) BAIL();

  /* children that are produced inside this list need no upload; everything else must be
   * current on the device (tips, CLVs computed by an earlier call and since edited on the host).
   * lvl_clv_r / lvl_sc_r are reused as "written by an earlier op of this list" flags. */
  for (i = 0; i < p->nodes; ++i) x->lvl_clv_r[i] = 0;
  for (i = 0; i < p->scale_buffers; ++i) x->lvl_sc_r[i] = 0;
  for (i = 0; i < count; ++i)
  {
    const pll_operation_t *o = &ops[i];
    const unsigned int kids[2] = {o->child1_clv_index, o->child2_clv_index};
    const int ksc[2] = {o->child1_scaler_index, o->child2_scaler_index};
    for (int c = 0; c < 2; ++c)
    {
      if (!x->lvl_clv_r[kids[c]] && !pll_flush_clv(p, x, kids[c])) BAIL();
      if (ksc[c] >= 0 && !pll_tip_by_codes(p, kids[c]) && !x->lvl_sc_r[ksc[c]] && !pll_flush_scaler(p, x, ksc[c])) BAIL();
      if (rep && !pll_flush_repeats(p, x, kids[c])) BAIL();
    }
    if (rep && !pll_flush_repeats(p, x, o->parent_clv_index)) BAIL();
    x->lvl_clv_r[o->parent_clv_index] = 1;
    if (o->parent_scaler_index >= 0) x->lvl_sc_r[o->parent_scaler_index] = 1;
  }

  /* 3. classify + order by level (stable counting sort) */
  unsigned int *start = (unsigned int *)calloc(nlevels + 1, sizeof(unsigned int));
  if (!start)
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_update_partials: out of memory");
    BAIL();
  }
  for (i = 0; i < count; ++i) start[level[i] + 1]++;
  for (i = 0; i < nlevels; ++i) start[i + 1] += start[i];
  for (i = 0; i < count; ++i)
  {
    const pll_operation_t *o = &ops[i];
    pllgpu_op_t *g = &x->gops[start[level[i]]++];
    const int t1 = pll_tip_by_codes(p, o->child1_clv_index);
    const int t2 = pll_tip_by_codes(p, o->child2_clv_index);
    /* the tip goes left in a tip-inner pair (src/partials.c:90-112) */
    const int swap = (!t1 && t2);
    g->parent_clv = o->parent_clv_index;
    g->parent_scaler = o->parent_scaler_index;
    g->left_clv = swap ? o->child2_clv_index : o->child1_clv_index;
    g->right_clv = swap ? o->child1_clv_index : o->child2_clv_index;
    g->left_matrix = swap ? o->child2_matrix_index : o->child1_matrix_index;
    g->right_matrix = swap ? o->child1_matrix_index : o->child2_matrix_index;
    g->left_scaler = swap ? o->child2_scaler_index : o->child1_scaler_index;
    g->right_scaler = swap ? o->child1_scaler_index : o->child2_scaler_index;
    g->parent_entries = pll_get_sites_number(p, o->parent_clv_index);
    g->flags = 0;
    if (t1 || t2) g->flags |= PLLGPU_OP_LEFT_TIP;
    if (t1 && t2) g->flags |= PLLGPU_OP_RIGHT_TIP;
    /* tips carry no scaler (src/parse_utree.y:271-336); a stray index on a tip is ignored as the
     * reference's tip kernels do (src/core_partials.c:282,463) */
    if (t1 || t2) g->left_scaler = PLL_SCALE_BUFFER_NONE;
    if (t1 && t2) g->right_scaler = PLL_SCALE_BUFFER_NONE;
    if (rep && (p->repeats->pernode_ids[o->parent_clv_index] || p->repeats->pernode_ids[o->child1_clv_index] ||
                p->repeats->pernode_ids[o->child2_clv_index]))
      g->flags |= PLLGPU_OP_GATHER;
    g->level = level[i];
    g->war_level = war[i];
  }
  free(start);

  /* 4. launch */
  if (pllgpu_update_partials(x->ctx, x->gops, count) != 0)
  {
    pll_set_gpu_error("pll_update_partials");
    return;
  }
  mark_results(p, x, ops, count);
  /* remember the list: the next call with the same one may skip everything above (nothing_dirty) */
  if (count > x->fast_cap)
  {
    free(x->fast_ops);
    x->fast_ops = (pll_operation_t *)malloc(sizeof(pll_operation_t) * count);
    x->fast_cap = x->fast_ops ? count : 0;
  }
  if (x->fast_ops)
  {
    memcpy(x->fast_ops, ops, sizeof(pll_operation_t) * count);
    x->fast_count = count;
    x->fast_lo = lo;
    x->fast_hi = hi;
    x->fast_valid = 1;
  }
  if (x->eager_mirror)
    for (i = 0; i < count; ++i)
    {
      pll_gpu_sync_clv(p, ops[i].parent_clv_index);
      if (ops[i].parent_scaler_index >= 0) pll_gpu_sync_scaler(p, (unsigned)ops[i].parent_scaler_index);
    }
}
