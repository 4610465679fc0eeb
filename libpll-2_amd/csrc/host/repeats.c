/* repeats.c - site-repeats bookkeeping.
 *
 * Semantics of src/repeats.c: a node's CLV holds one entry per CLASS of sites that are
 * indistinguishable in the subtree below it. pernode_site_id[node][site] is the class of a site,
 * pernode_id_site[node][class] its first site; classes are numbered by first occurrence. A
 * parent's classes are the distinct (left class, right class) pairs, found with a direct-address
 * table that is wiped after use through a to-clean list (src/repeats.c:334-377).
 *
 * With a device context the class maps of inner nodes are computed ON the device
 * (pll_update_repeats_device -> kernels_repeats.h), all dependency levels of a list in one call; the
 * host learns the class counts once, afterwards (its allocation sizes need them; the decision whether
 * a parent is compressed is taken on the device by the reference's default rule - a caller-supplied
 * enable_repeats callback is asked level by level instead) and
 * partition->repeats->pernode_site_id / pernode_id_site become a mirror that
 * pll_get_site_id / pll_get_id_site / pll_gpu_sync_repeats refresh on demand. Tip maps are built on
 * the host while the sequence is parsed and uploaded. The sequential table walk below
 * (pll_update_repeats_host) only serves host-only shells (PLL_AMD_HOST_ONLY, CPU tests).
 */
#include "pll_internal.h"

#define EMPTY 0xFFFFFFFFu

int pll_repeats_enabled(const pll_partition_t *p) { return (p->attributes & PLL_ATTRIB_SITE_REPEATS) != 0; }

void pll_resize_repeats_lookup(pll_partition_t *p, unsigned int size)
{
  { pll_amd_ext_t *xf_ = p ? pll_ext(p) : NULL; if (xf_) xf_->fast_valid = 0; } /* class counts may change: partials.c's fast path starts over */
  if (!size) return;
  pll_repeats_t *r = p->repeats;
  free(r->lookup_buffer);
  r->lookup_buffer_size = size;
  r->lookup_buffer = (unsigned int *)malloc((size_t)size * sizeof(unsigned int));
  memset(r->lookup_buffer, 0xFF, (size_t)size * sizeof(unsigned int));
}

unsigned int pll_get_sites_number(const pll_partition_t *p, unsigned int clv_index)
{
  unsigned int n = pll_repeats_enabled(p) ? p->repeats->pernode_ids[clv_index] : 0;
  if (!n) n = p->sites;
  return n + (p->asc_bias_alloc ? p->states : 0);
}

unsigned int pll_get_clv_size(const pll_partition_t *p, unsigned int clv_index)
{
  return pll_get_sites_number(p, clv_index) * p->states_padded * p->rate_cats;
}

/* the accessors hand out host pointers: bring the mirror up to date first */
static void mirror_maps(const pll_partition_t *p, unsigned int clv_index)
{
  const pll_amd_ext_t *x = pll_ext(p);
  if (x && x->ctx && x->repeats_stale[clv_index]) (void)pll_gpu_sync_repeats((pll_partition_t *)p, (int)clv_index);
}

unsigned int *pll_get_site_id(const pll_partition_t *p, unsigned int clv_index)
{
  if (pll_repeats_enabled(p) && p->repeats->pernode_ids[clv_index])
  {
    mirror_maps(p, clv_index);
    return p->repeats->pernode_site_id[clv_index];
  }
  return NULL;
}

unsigned int *pll_get_id_site(const pll_partition_t *p, unsigned int clv_index)
{
  if (pll_repeats_enabled(p) && p->repeats->pernode_ids[clv_index])
  {
    mirror_maps(p, clv_index);
    return p->repeats->pernode_id_site[clv_index];
  }
  return NULL;
}

unsigned int pll_default_enable_repeats(pll_partition_t *p, unsigned int left, unsigned int right)
{
  /* src/repeats.c:100-110: compress only if the pair table fits and both children are
   * themselves compressed to at most half the sites */
  const pll_repeats_t *r = p->repeats;
  const unsigned long long cells = (unsigned long long)r->pernode_ids[left] * r->pernode_ids[right];
  if (!cells || cells >= r->lookup_buffer_size) return 0;
  if (r->pernode_ids[left] > p->sites / 2 || r->pernode_ids[right] > p->sites / 2) return 0;
  return 1;
}

unsigned int pll_no_enable_repeats(pll_partition_t *p, unsigned int left, unsigned int right)
{
  (void)p;
  (void)left;
  (void)right;
  return 0;
}

int pll_repeats_initialize(pll_partition_t *p)
{
  { pll_amd_ext_t *xf_ = p ? pll_ext(p) : NULL; if (xf_) xf_->fast_valid = 0; } /* class counts may change: partials.c's fast path starts over */
  const unsigned int n = pll_sites_alloc(p);
  unsigned int i;
  pll_repeats_t *r = (pll_repeats_t *)calloc(1, sizeof(pll_repeats_t));
  p->repeats = r;
  if (!r) goto oom;
  r->enable_repeats = pll_default_enable_repeats;
  r->reallocate_repeats = pll_default_reallocate_repeats;
  r->pernode_site_id = (unsigned int **)calloc(p->nodes, sizeof(unsigned int *));
  r->pernode_id_site = (unsigned int **)calloc(p->nodes, sizeof(unsigned int *));
  if (!r->pernode_site_id || !r->pernode_id_site) goto oom;
  for (i = 0; i < p->nodes; ++i)
  {
    r->pernode_site_id[i] = (unsigned int *)calloc(n, sizeof(unsigned int));
    r->pernode_id_site[i] = (unsigned int *)calloc(n, sizeof(unsigned int));
    if (!r->pernode_site_id[i] || !r->pernode_id_site[i]) goto oom;
  }
  r->pernode_ids = (unsigned int *)calloc(p->nodes, sizeof(unsigned int));
  r->perscale_ids = (unsigned int *)calloc(p->scale_buffers ? p->scale_buffers : 1, sizeof(unsigned int));
  r->pernode_allocated_clvs = (unsigned int *)calloc(p->nodes, sizeof(unsigned int));
  r->toclean_buffer = (unsigned int *)malloc(n * sizeof(unsigned int));
  r->id_site_buffer = (unsigned int *)malloc(n * sizeof(unsigned int));
  /* bclv_buffer: the reference's scratch for a precomputed left term; the device kernels do not
   * use it, a small non-NULL block keeps pll_disable_bclv and callers' NULL tests meaningful */
  r->bclv_buffer = (double *)pll_aligned_alloc(64, p->alignment);
  r->charmap = (char *)calloc(PLL_ASCII_SIZE, 1);
  if (!r->pernode_ids || !r->perscale_ids || !r->pernode_allocated_clvs || !r->toclean_buffer ||
      !r->id_site_buffer || !r->bclv_buffer || !r->charmap)
    goto oom;
  return PLL_SUCCESS;
oom:
  pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory for repeats structure.");
  return PLL_FAILURE;
}

void pll_disable_bclv(pll_partition_t *p)
{
  if (!pll_repeats_enabled(p)) return;
  free(p->repeats->bclv_buffer);
  p->repeats->bclv_buffer = NULL;
}

/* ---- which class maps still stand (pll_internal.h: map_version / map_stamp) ------------------------------------
 * The reference recomputes a parent's map on every pll_update_partials (src/partials.c:253-255) although the map is a
 * function of the two children's maps and the enable rule only. Here every write of a node's map gives the node a new
 * version, and a parent remembers the (child, version) pairs it was computed from: while they are what they were the
 * parent's map - and its version - stand, so the question "did anything move" costs a comparison per op and an
 * unchanged tree costs no launch at all. */
void pll_maps_touched(pll_amd_ext_t *x, const pll_partition_t *p, int node)
{
  unsigned int i;
  if (!x || !x->map_version) return;
  for (i = 0; i < p->nodes; ++i)
    if (node < 0 || i == (unsigned int)node)
    {
      x->map_version[i] = ++x->map_clock;
      x->map_stamp[i].valid = 0;
    }
}

static int stamp_holds(const pll_amd_ext_t *x, const pll_repeats_t *r, const pll_operation_t *op)
{
  const pll_map_stamp_t *s = &x->map_stamp[op->parent_clv_index];
  return s->valid && s->left == op->child1_clv_index && s->right == op->child2_clv_index &&
         s->lver == x->map_version[op->child1_clv_index] && s->rver == x->map_version[op->child2_clv_index] &&
         s->lookup == r->lookup_buffer_size;
}

/* the parent's map is about to be computed from the children's maps as they are now */
static void stamp_set(pll_amd_ext_t *x, const pll_repeats_t *r, const pll_operation_t *op)
{
  pll_map_stamp_t *s = &x->map_stamp[op->parent_clv_index];
  s->left = op->child1_clv_index;
  s->right = op->child2_clv_index;
  s->lver = x->map_version[op->child1_clv_index];
  s->rver = x->map_version[op->child2_clv_index];
  s->lookup = r->lookup_buffer_size;
  s->valid = 1;
  x->map_version[op->parent_clv_index] = ++x->map_clock;
}

/* classes of a tip = distinct state masks of its sequence (src/repeats.c:189-254) */
int pll_update_repeats_tips(pll_partition_t *p, unsigned int tip, const pll_state_t *map, const char *seq)
{
  { pll_amd_ext_t *xf_ = p ? pll_ext(p) : NULL; if (xf_) xf_->fast_valid = 0; } /* class counts may change: partials.c's fast path starts over */
  pll_repeats_t *r = p->repeats;
  unsigned int i, j, s, next = 0;
  if (!r->lookup_buffer) pll_resize_repeats_lookup(p, PLL_REPEATS_LOOKUP_SIZE);

  /* characters with equal masks must fall into one class: number the distinct masks 1,2,.. */
  char label = 0;
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
  {
    for (j = 0; j < i; ++j)
      if (map[i] == map[j])
      {
        r->charmap[i] = r->charmap[j];
        break;
      }
    if (!r->charmap[i]) r->charmap[i] = ++label;
  }

  unsigned int *site_id = r->pernode_site_id[tip];
  for (s = 0; s < p->sites; ++s)
  {
    const unsigned int cell = (unsigned int)r->charmap[(unsigned char)seq[s]];
    if (r->lookup_buffer[cell] == EMPTY)
    {
      r->toclean_buffer[next] = cell;
      r->id_site_buffer[next] = s;
      r->lookup_buffer[cell] = next++;
    }
    site_id[s] = r->lookup_buffer[cell];
  }
  r->pernode_ids[tip] = next;
  free(r->pernode_id_site[tip]);
  r->pernode_id_site[tip] = (unsigned int *)malloc((next ? next : 1) * sizeof(unsigned int));
  for (s = 0; s < next; ++s)
  {
    r->pernode_id_site[tip][s] = r->id_site_buffer[s];
    r->lookup_buffer[r->toclean_buffer[s]] = EMPTY;
  }
  const size_t bytes = (size_t)next * p->states_padded * p->rate_cats * sizeof(double);
  free(p->clv[tip]);
  p->clv[tip] = (double *)pll_aligned_alloc(bytes ? bytes : 8, p->alignment);
  if (!p->clv[tip])
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory for repeats structure.");
    return PLL_FAILURE;
  }
  memset(p->clv[tip], 0, bytes);
  r->pernode_allocated_clvs[tip] = next;
  pll_amd_ext_t *x = pll_ext(p);
  if (x)
  {
    x->repeats_dirty[tip] = 1;
    x->repeats_stale[tip] = 0;
    x->repeats_count[tip] = next;
    pll_maps_touched(x, p, (int)tip);
  }
  return PLL_SUCCESS;
}

/* host mirror sizing for a parent whose class count changed (src/repeats.c:256-296). The device
 * buffers are sized by the launch code from the op's entry count. */
void pll_default_reallocate_repeats(pll_partition_t *p, unsigned int parent, int scaler_index,
                                    unsigned int sites_to_alloc)
{
  { pll_amd_ext_t *xf_ = p ? pll_ext(p) : NULL; if (xf_) xf_->fast_valid = 0; }
  pll_repeats_t *r = p->repeats;
  if (sites_to_alloc == r->pernode_allocated_clvs[parent]) return;
  r->pernode_allocated_clvs[parent] = sites_to_alloc;
  free(p->clv[parent]);
  const size_t bytes = (size_t)sites_to_alloc * p->states_padded * p->rate_cats * sizeof(double);
  p->clv[parent] = (double *)pll_aligned_alloc(bytes ? bytes : 8, p->alignment);
  if (!p->clv[parent])
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory for repeats structure.");
    return;
  }
  if (scaler_index != PLL_SCALE_BUFFER_NONE)
  {
    size_t n = sites_to_alloc;
    if (p->attributes & PLL_ATTRIB_RATE_SCALERS) n *= p->rate_cats;
    free(p->scale_buffer[scaler_index]);
    p->scale_buffer[scaler_index] = (unsigned int *)calloc(n ? n : 1, sizeof(unsigned int));
  }
  free(r->pernode_id_site[parent]);
  r->pernode_id_site[parent] = (unsigned int *)malloc((sites_to_alloc ? sites_to_alloc : 1) * sizeof(unsigned int));
}

/* what follows a parent's class count, on either path (src/repeats.c:349-381) */
static void adopt_classes(pll_partition_t *p, const pll_operation_t *op, unsigned int classes, int enabled)
{
  pll_repeats_t *r = p->repeats;
  const unsigned int parent = op->parent_clv_index;
  const unsigned int to_alloc = enabled ? classes : p->sites;
  r->pernode_ids[parent] = enabled ? classes : 0;
  if (op->parent_scaler_index != PLL_SCALE_BUFFER_NONE) r->perscale_ids[op->parent_scaler_index] = enabled ? classes : 0;
  r->reallocate_repeats(p, parent, op->parent_scaler_index, to_alloc);
  /* no compression gained: fall back to one entry per site (:364-370) */
  if (to_alloc >= p->sites)
  {
    r->pernode_ids[parent] = 0;
    if (op->parent_scaler_index != PLL_SCALE_BUFFER_NONE) r->perscale_ids[op->parent_scaler_index] = 0;
  }
}

/* what follows an op's count word (PLLGPU_REPEATS_COMPRESSED | classes, or 0: the parent stays uncompressed) on the host */
static int follow_count(pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *op, unsigned int word, int *changed)
{
  pll_repeats_t *r = p->repeats;
  const unsigned int parent = op->parent_clv_index;
  const int enabled = (word & PLLGPU_REPEATS_COMPRESSED) != 0;
  const unsigned int classes = word & ~PLLGPU_REPEATS_COMPRESSED;
  /* the same classes as before (a re-evaluation of the same tree): what pll_update_partials classified stays right,
   * and with the default callback nothing below would change a thing - the host's time between the counts and the
   * first launch that uses the maps is the device's idle time on a small alignment */
  const int same = x->repeats_count[parent] == (enabled ? classes : 0) && r->pernode_ids[parent] == (enabled && classes < p->sites ? classes : 0) &&
                   r->pernode_allocated_clvs[parent] == (enabled ? classes : p->sites);
  x->repeats_stale[parent] = enabled ? 1 : 0;
  if (same && r->reallocate_repeats == pll_default_reallocate_repeats)
  {
    if (op->parent_scaler_index != PLL_SCALE_BUFFER_NONE) r->perscale_ids[op->parent_scaler_index] = r->pernode_ids[parent];
    return PLL_SUCCESS;
  }
  if (!same) *changed = 1;
  adopt_classes(p, op, classes, enabled);
  x->repeats_count[parent] = enabled ? classes : 0;
  if (pllgpu_repeats_set_ids(x->ctx, parent, r->pernode_ids[parent]) != 0)
  {
    pll_set_gpu_error("pll_update_repeats");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

/* one piece of work for pllgpu_repeats_classes and what follows its counts */
static int classes_call(pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *ops, pllgpu_repop_t *rop,
                        const unsigned int *idx, unsigned int n, unsigned int *counts, int *changed)
{
  unsigned int k;
  if (!n) return PLL_SUCCESS;
  if (pllgpu_repeats_classes(x->ctx, rop, n, p->repeats->lookup_buffer_size, counts) != 0)
  {
    pll_set_gpu_error("pll_update_repeats");
    return PLL_FAILURE;
  }
  for (k = 0; k < n; ++k)
    if (!follow_count(p, x, &ops[idx[k]], counts[k], changed)) return PLL_FAILURE;
  return PLL_SUCCESS;
}

/* Class maps of an op list on the device. With the reference's own decision rule (pll_default_enable_repeats) the
 * whole list goes down in one call: the rule is a function of the children's class counts, which the device holds
 * before the host does, so the levels run back to back and the host reads all counts once (round 4: a blocking
 * hand-off per level - the callback wanted the counts on the host). A caller-supplied enable_repeats keeps that form:
 * the callback is asked level by level, with the counts of the levels below in pernode_ids as the reference has them.
 * PLL_AMD_REP_LEVEL_SYNC=1 takes the level-by-level form for the default rule too (A/B, tests). */
int pll_update_repeats_device(pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *ops,
                              unsigned int count, const unsigned int *level, unsigned int nlevels)
{
  const int was_fast = x->fast_valid;
  int changed = 0;
  x->fast_valid = 0;
  pll_repeats_t *r = p->repeats;
  unsigned int l, i, n;
  int ok = PLL_FAILURE;
  if (!r->lookup_buffer)
  {
    pll_resize_repeats_lookup(p, PLL_REPEATS_LOOKUP_SIZE); /* its SIZE bounds the pair table */
    changed = 1;
  }
  pllgpu_repop_t *rop = (pllgpu_repop_t *)malloc(sizeof(pllgpu_repop_t) * count);
  unsigned int *idx = (unsigned int *)malloc(sizeof(unsigned int) * 2 * count);
  int *producer = (int *)malloc(sizeof(int) * (p->nodes ? p->nodes : 1));
  if (!rop || !idx || !producer)
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory for repeats structure.");
    goto done;
  }
  unsigned int *counts = idx + count;
  static int level_sync = -1;
  if (level_sync < 0)
  {
    const char *v = getenv("PLL_AMD_REP_LEVEL_SYNC");
    level_sync = (v && *v && *v != '0') ? 1 : 0;
  }
  if (r->enable_repeats == pll_default_enable_repeats && !level_sync && count <= PLLGPU_REPEATS_MAX_OPS)
  {
    /* which ops have anything to compute: in list order, so that a parent sees the versions its children's maps have
     * when ITS turn comes (a child written by an earlier op of the list that is computed again has a new one) */
    unsigned char *keep = (unsigned char *)malloc(count);
    unsigned int *start = (unsigned int *)calloc(nlevels + 1, sizeof(unsigned int));
    unsigned int *pos = (unsigned int *)malloc(sizeof(unsigned int) * count); /* op of the list -> its place in the call */
    if (!keep || !start || !pos)
    {
      free(keep);
      free(start);
      free(pos);
      pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory for repeats structure.");
      goto done;
    }
    const int stamps = x->map_stamps && r->reallocate_repeats == pll_default_reallocate_repeats;
    /* ... and which of those cannot be compressed whatever the device finds: the rule (src/repeats.c:100-110) wants
     * both children compressed, so a parent over a child that is not - known here for a child from outside the call, and
     * from there up the list - stays uncompressed: no launch, the host's bookkeeping alone. In a tree search that is most of
     * a partial traversal (the path from the moved edge to the evaluated one runs through large subtrees):
     * keep = 1: to the device, 2: settled here. producer[]: -1 outside the call, -2 settled here, else the place in the call */
    unsigned int kept = 0;
    for (i = 0; i < p->nodes; ++i) producer[i] = -1;
    for (i = 0; i < count; ++i)
    {
      keep[i] = !(stamps && stamp_holds(x, r, &ops[i]));
      if (!keep[i]) continue;
      stamp_set(x, r, &ops[i]);
      const unsigned int left = ops[i].child1_clv_index, right = ops[i].child2_clv_index;
      if (producer[left] == -2 || producer[right] == -2 || (producer[left] == -1 && !r->pernode_ids[left]) ||
          (producer[right] == -1 && !r->pernode_ids[right]))
      {
        keep[i] = 2;
        producer[ops[i].parent_clv_index] = -2;
        continue;
      }
      producer[ops[i].parent_clv_index] = 0; /* (its place follows) */
      start[level[i] + 1]++;
      ++kept;
    }
    /* order by level (stable); a child's producer is the latest earlier op of the CALL that writes it */
    for (l = 0; l < nlevels; ++l) start[l + 1] += start[l];
    for (i = 0; i < count; ++i)
      if (keep[i] == 1) pos[i] = start[level[i]]++;
    for (i = 0; i < p->nodes; ++i) producer[i] = -1;
    int failed = 0;
    for (i = 0; i < count && !failed; ++i)
    {
      if (!keep[i]) continue;
      const pll_operation_t *op = &ops[i];
      const unsigned int left = op->child1_clv_index, right = op->child2_clv_index, parent = op->parent_clv_index;
      x->repeats_dirty[parent] = 0;
      if (keep[i] == 2)
      {
        producer[parent] = -1; /* (nothing of the call reads a settled parent's map: whoever does is settled too) */
        continue;
      }
      pllgpu_repop_t *o = &rop[pos[i]];
      o->parent = parent;
      o->left = left;
      o->right = right;
      o->lsrc = producer[left];
      o->rsrc = producer[right];
      o->nleft = o->lsrc >= 0 ? 0 : r->pernode_ids[left];
      o->nright = o->rsrc >= 0 ? 0 : r->pernode_ids[right];
      o->level = level[i];
      o->force = 0;
      idx[pos[i]] = i;
      /* tips: host-built maps go up now */
      if (o->lsrc < 0 && o->nleft && !pll_flush_repeats(p, x, left)) failed = 1;
      if (o->rsrc < 0 && o->nright && !pll_flush_repeats(p, x, right)) failed = 1;
      producer[parent] = (int)pos[i];
    }
    free(start);
    free(pos);
    if (!failed) ok = classes_call(p, x, ops, rop, idx, kept, counts, &changed);
    for (i = 0; i < count && ok; ++i)
      if (keep[i] == 2) ok = follow_count(p, x, &ops[i], 0u, &changed);
    if (!ok) /* whatever the device left of these maps is not what their stamps say */
      for (i = 0; i < count; ++i)
        if (keep[i]) pll_maps_touched(x, p, (int)ops[i].parent_clv_index);
    free(keep);
    if (failed) goto done;
  }
  else
  {
    ok = PLL_SUCCESS;
    /* level by level with the decisions on the host: no stamps kept (a caller's rule may depend on anything) */
    for (i = 0; i < count; ++i) pll_maps_touched(x, p, (int)ops[i].parent_clv_index);
    for (l = 0; l < nlevels && ok; ++l)
    {
      n = 0;
      for (i = 0; i < count && ok; ++i)
      {
        if (level[i] != l) continue;
        const pll_operation_t *op = &ops[i];
        const unsigned int left = op->child1_clv_index, right = op->child2_clv_index, parent = op->parent_clv_index;
        x->repeats_dirty[parent] = 0;
        if (!r->enable_repeats(p, left, right))
        {
          if (x->repeats_count[parent] || r->pernode_ids[parent] || r->pernode_allocated_clvs[parent] != p->sites) changed = 1;
          adopt_classes(p, op, 0, 0);
          x->repeats_stale[parent] = 0;
          x->repeats_count[parent] = 0;
          if (pllgpu_repeats_set_ids(x->ctx, parent, 0) != 0) goto gpu_fail;
          continue;
        }
        /* tips: host-built maps go up now; inner children were produced by an earlier level */
        if (!pll_flush_repeats(p, x, left) || !pll_flush_repeats(p, x, right))
        {
          ok = PLL_FAILURE;
          break;
        }
        rop[n].parent = parent;
        rop[n].left = left;
        rop[n].right = right;
        rop[n].nleft = r->pernode_ids[left];
        rop[n].nright = r->pernode_ids[right];
        rop[n].lsrc = rop[n].rsrc = -1;
        rop[n].level = 0;
        rop[n].force = 1;
        idx[n++] = i;
        if (n == PLLGPU_REPEATS_MAX_OPS)
        {
          ok = classes_call(p, x, ops, rop, idx, n, counts, &changed);
          n = 0;
        }
      }
      if (ok) ok = classes_call(p, x, ops, rop, idx, n, counts, &changed);
    }
  }
  if (ok && x->eager_mirror) ok = pll_gpu_sync_repeats(p, -1);
  /* every class count as it was: the classified list of the last whole pll_update_partials (partials.c: fast path) is
   * still the right one */
  if (ok && !changed) x->fast_valid = was_fast;
  goto done;
gpu_fail:
  pll_set_gpu_error("pll_update_repeats");
  ok = PLL_FAILURE;
done:
  free(rop);
  free(idx);
  free(producer);
  return ok;
}

static void pll_update_repeats_host(pll_partition_t *p, const pll_operation_t *op);

void pll_update_repeats(pll_partition_t *p, const pll_operation_t *op)
{
  { pll_amd_ext_t *xf_ = p ? pll_ext(p) : NULL; if (xf_) xf_->fast_valid = 0; } /* class counts may change: partials.c's fast path starts over */
  pll_amd_ext_t *x = pll_ext(p);
  if (x && x->ctx)
  {
    const unsigned int level = 0;
    (void)pll_update_repeats_device(p, x, op, 1, &level, 1);
    return;
  }
  pll_update_repeats_host(p, op);
}

static void pll_update_repeats_host(pll_partition_t *p, const pll_operation_t *op)
{
  pll_repeats_t *r = p->repeats;
  const unsigned int left = op->child1_clv_index, right = op->child2_clv_index, parent = op->parent_clv_index;
  unsigned int s, classes = 0, to_alloc;
  if (!r->lookup_buffer) pll_resize_repeats_lookup(p, PLL_REPEATS_LOOKUP_SIZE);

  if (!r->enable_repeats(p, left, right))
  {
    to_alloc = p->sites;
    r->pernode_ids[parent] = 0;
    if (op->parent_scaler_index != PLL_SCALE_BUFFER_NONE) r->perscale_ids[op->parent_scaler_index] = 0;
  }
  else
  {
    const unsigned int *lid = r->pernode_site_id[left];
    const unsigned int *rid = r->pernode_site_id[right];
    unsigned int *pid = r->pernode_site_id[parent];
    const unsigned int nleft = r->pernode_ids[left];
    for (s = 0; s < p->sites; ++s)
    {
      const unsigned int cell = lid[s] + rid[s] * nleft;
      unsigned int id = r->lookup_buffer[cell];
      if (id == EMPTY)
      {
        r->toclean_buffer[classes] = cell;
        r->id_site_buffer[classes] = s;
        id = classes;
        r->lookup_buffer[cell] = classes++;
      }
      pid[s] = id;
    }
    r->pernode_ids[parent] = classes;
    if (op->parent_scaler_index != PLL_SCALE_BUFFER_NONE) r->perscale_ids[op->parent_scaler_index] = classes;
    to_alloc = classes;
  }

  r->reallocate_repeats(p, parent, op->parent_scaler_index, to_alloc);

  /* no compression gained: fall back to one entry per site (src/repeats.c:364-370) */
  if (to_alloc >= p->sites)
  {
    r->pernode_ids[parent] = 0;
    if (op->parent_scaler_index != PLL_SCALE_BUFFER_NONE) r->perscale_ids[op->parent_scaler_index] = 0;
  }
  for (s = 0; s < classes; ++s)
  {
    r->pernode_id_site[parent][s] = r->id_site_buffer[s];
    r->lookup_buffer[r->toclean_buffer[s]] = EMPTY;
  }
  pll_amd_ext_t *x = pll_ext(p);
  if (x)
  {
    x->repeats_dirty[parent] = 1;
    pll_maps_touched(x, p, (int)parent);
  }
}

/* ---- scaler utilities for class-compressed operands (src/repeats.c:392-540) ----------------------
 * parent entry i stands for site psites[i] (identity without the map), whose children's entries are
 * lids[site] / rids[site] (identity without a map); a missing scaler contributes nothing. */
void pll_fill_parent_scaler_repeats_per_rate(unsigned int sites, unsigned int rates, unsigned int *parent_scaler,
                                             const unsigned int *psites, const unsigned int *left_scaler, const unsigned int *lids,
                                             const unsigned int *right_scaler, const unsigned int *rids)
{
  for (unsigned int i = 0; i < sites; ++i)
  {
    const unsigned int site = psites ? psites[i] : i;
    const unsigned int l = lids ? lids[site] : site, r = rids ? rids[site] : site;
    for (unsigned int k = 0; k < rates; ++k)
      parent_scaler[(size_t)i * rates + k] = (left_scaler ? left_scaler[(size_t)l * rates + k] : 0u) +
                                             (right_scaler ? right_scaler[(size_t)r * rates + k] : 0u);
  }
}

void pll_fill_parent_scaler_repeats(unsigned int sites, unsigned int *parent_scaler, const unsigned int *psites,
                                    const unsigned int *left_scaler, const unsigned int *lids, const unsigned int *right_scaler,
                                    const unsigned int *rids)
{
  pll_fill_parent_scaler_repeats_per_rate(sites, 1, parent_scaler, psites, left_scaler, lids, right_scaler, rids);
}
