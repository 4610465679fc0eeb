// fusion_plan.h - part of the pllgpu.hip translation unit (included there, after the context and the
// launch helpers): which ops of a level-scheduled list are evaluated together (three-op groups,
// seven-op cherry-cherry groups; kernels_dna.h), and the launches of those groups.
#pragma once
// ---- producer/consumer fusion (DNA) ---------------------------------------------------------------
// A group = an op P plus the ops that produce its children in the SAME call, evaluated by one
// kernel at the producers' level (kernels_dna.h: k_partials_dna_fused). P may move one level up
// only if nothing it must wait for sits at that level: its non-fused child is older, and no earlier
// op of the list still reads or writes P's outputs there (war_level, from the host's scheduler).
struct FusedGroup
{
  unsigned p;
  int a, b;       // producer ops of the left / right child, or -1
  int lk, rk;     // DnaChildKind of the left / right child
  unsigned level; // execution level (= level of the lowest producers)
  int aa = -1, ab = -1, ba = -1, bb = -1; // CK_FCC sides: the cherries under a / b
  int ga = -1, gb = -1;                   // CK_F8 sides: the complete 8-tip groups (indices into the group list) whose parents are a / b
  bool absorbed = false;                  // a complete 8-tip group that a fifteen-op group evaluates: not launched on its own
};

// the group list ordered by level (stable), with the group indices some groups hold (ga / gb) following their groups
static void sort_groups_by_level(std::vector<FusedGroup> &groups)
{
  if (std::is_sorted(groups.begin(), groups.end(), [](const FusedGroup &x, const FusedGroup &y) { return x.level < y.level; })) return;
  std::vector<size_t> order(groups.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return groups[x].level < groups[y].level; });
  std::vector<int> now_at(groups.size());
  std::vector<FusedGroup> sorted(groups.size());
  for (size_t i = 0; i < order.size(); ++i)
  {
    now_at[order[i]] = (int)i;
    sorted[i] = groups[order[i]];
  }
  for (FusedGroup &g : sorted)
  {
    if (g.ga >= 0) g.ga = now_at[g.ga];
    if (g.gb >= 0) g.gb = now_at[g.gb];
  }
  groups.swap(sorted);
}

static int child_kind(const pllgpu_op_t &prod)
{
  const bool lt = prod.flags & PLLGPU_OP_LEFT_TIP, rt = prod.flags & PLLGPU_OP_RIGHT_TIP;
  return (lt && rt) ? CK_FTT : lt ? CK_FTI : CK_FII;
}

static void plan_fusion(bool fuse, bool fuse_cc, unsigned nodes, const pllgpu_op_t *ops, unsigned count, std::vector<int> &role,
                        std::vector<FusedGroup> &groups, bool cc_only = false, bool cc16 = false)
{
  // role: 0 plain, 1 parent of a group, 2 fused into a group as a child
  role.assign(count, 0);
  groups.clear();
  if (!fuse) return;
  std::vector<int> producer(nodes, -1), prod_l(count, -1), prod_r(count, -1);
  std::vector<unsigned> eff(count);
  // ops arrive sorted by level; same-level ops are independent, so the producer table may be
  // updated as we go
  for (unsigned i = 0; i < count; ++i)
  {
    eff[i] = ops[i].level;
    if (!(ops[i].flags & PLLGPU_OP_LEFT_TIP)) prod_l[i] = producer[ops[i].left_clv];
    if (!(ops[i].flags & PLLGPU_OP_RIGHT_TIP)) prod_r[i] = producer[ops[i].right_clv];
    producer[ops[i].parent_clv] = (int)i;
  }
  // first pass: parents of complete cherry-cherry subtrees (kind CK_FCC, kernels_dna.h): P two levels
  // above the cherries, its child A an inner x inner op between them. Everything moves to the
  // cherries' level, so neither A's nor P's outputs may be touched by an earlier op from there on.
  if (fuse_cc)
    for (unsigned i = 0; i < count; ++i)
    {
      const pllgpu_op_t &P = ops[i];
      if (role[i] || (P.flags & PLLGPU_OP_GATHER) || P.level < 2) continue;
      const unsigned L = P.level - 2;
      if (P.war_level >= (int)L) continue;
      auto cherry = [&](int pr, int pscal, unsigned entries) {
        return pr >= 0 && role[pr] == 0 && ops[pr].level == L && !(ops[pr].flags & PLLGPU_OP_GATHER) &&
               (ops[pr].flags & PLLGPU_OP_LEFT_TIP) && (ops[pr].flags & PLLGPU_OP_RIGHT_TIP) && ops[pr].parent_scaler == pscal &&
               ops[pr].parent_entries == entries;
      };
      auto cc = [&](int pr, int pscal) {
        if (pr < 0 || role[pr] != 0 || ops[pr].level != L + 1 || (ops[pr].flags & (PLLGPU_OP_GATHER | PLLGPU_OP_LEFT_TIP | PLLGPU_OP_RIGHT_TIP)))
          return false;
        const pllgpu_op_t &A = ops[pr];
        return A.parent_scaler == pscal && A.parent_entries == P.parent_entries && A.war_level < (int)L &&
               prod_l[pr] != prod_r[pr] && cherry(prod_l[pr], A.left_scaler, P.parent_entries) && cherry(prod_r[pr], A.right_scaler, P.parent_entries);
      };
      auto memory_side = [&](int pr, bool tip) { return tip || pr < 0 || eff[pr] < L; };
      const bool ltip = P.flags & PLLGPU_OP_LEFT_TIP, rtip = P.flags & PLLGPU_OP_RIGHT_TIP;
      const bool cl = !ltip && cc(prod_l[i], P.left_scaler), cr = !rtip && prod_r[i] != prod_l[i] && cc(prod_r[i], P.right_scaler);
      if (!cl && !cr) continue;
      if ((!cl && !memory_side(prod_l[i], ltip)) || (!cr && !memory_side(prod_r[i], rtip))) continue;
      FusedGroup g;
      g.p = i;
      g.a = cl ? prod_l[i] : -1;
      g.b = cr ? prod_r[i] : -1;
      g.lk = cl ? CK_FCC : ltip ? CK_TIP : CK_INNER;
      g.rk = cr ? CK_FCC : rtip ? CK_TIP : CK_INNER;
      g.level = L;
      role[i] = 1;
      eff[i] = L;
      if (cl)
      {
        g.aa = prod_l[g.a];
        g.ab = prod_r[g.a];
        role[g.a] = role[g.aa] = role[g.ab] = 2;
      }
      if (cr)
      {
        g.ba = prod_l[g.b];
        g.bb = prod_r[g.b];
        role[g.b] = role[g.ba] = role[g.bb] = 2;
      }
      groups.push_back(g);
    }
  // second pass (chain plans only, kernels_dna.h: k_partials_dna_cc16): a parent three levels above the cherries whose
  // children are the parents of two COMPLETE 8-tip groups takes both with it. The groups' own conditions (scalers,
  // entries, nothing touching their outputs from the cherries' level on) were checked above; the same for P here.
  if (fuse_cc && cc16)
  {
    std::vector<int> group_of(count, -1);
    const size_t n8 = groups.size();
    for (size_t gi = 0; gi < n8; ++gi) group_of[groups[gi].p] = (int)gi;
    for (unsigned i = 0; i < count; ++i)
    {
      const pllgpu_op_t &P = ops[i];
      if (role[i] || (P.flags & (PLLGPU_OP_GATHER | PLLGPU_OP_LEFT_TIP | PLLGPU_OP_RIGHT_TIP)) || P.level < 3) continue;
      const unsigned L = P.level - 3;
      if (P.war_level >= (int)L) continue;
      const int pa = prod_l[i], pb = prod_r[i];
      if (pa < 0 || pb < 0 || pa == pb) continue;
      const int ga = group_of[pa], gb = group_of[pb];
      if (ga < 0 || gb < 0) continue;
      auto full8 = [&](const FusedGroup &g) { return g.lk == CK_FCC && g.rk == CK_FCC && g.level == L && !g.absorbed; };
      if (!full8(groups[ga]) || !full8(groups[gb])) continue;
      if (ops[pa].parent_scaler != P.left_scaler || ops[pb].parent_scaler != P.right_scaler) continue;
      if (ops[pa].parent_entries != P.parent_entries || ops[pb].parent_entries != P.parent_entries) continue;
      FusedGroup g;
      g.p = i;
      g.a = pa;
      g.b = pb;
      g.lk = g.rk = CK_F8;
      g.level = L;
      g.ga = ga;
      g.gb = gb;
      groups[ga].absorbed = groups[gb].absorbed = true;
      role[i] = 1;
      role[pa] = role[pb] = 2;
      eff[i] = L;
      groups.push_back(g);
    }
  }
  if (cc_only) return;
  for (unsigned i = 0; i < count; ++i)
  {
    const pllgpu_op_t &P = ops[i];
    if (role[i] || (P.flags & PLLGPU_OP_GATHER) || P.level == 0) continue;
    const unsigned L = P.level - 1;
    if (P.war_level >= (int)L) continue;
    auto fusable = [&](int pr, int pscal) {
      return pr >= 0 && role[pr] == 0 && eff[pr] == L && ops[pr].level == L && !(ops[pr].flags & PLLGPU_OP_GATHER) &&
             ops[pr].parent_scaler == pscal && ops[pr].parent_entries == P.parent_entries;
    };
    auto older = [&](int pr) { return pr < 0 || eff[pr] < L; };
    const bool fl = fusable(prod_l[i], P.left_scaler), fr = fusable(prod_r[i], P.right_scaler) && prod_r[i] != prod_l[i];
    if (!fl && !fr) continue;
    if ((!fl && !older(prod_l[i])) || (!fr && !older(prod_r[i]))) continue;
    FusedGroup g;
    g.p = i;
    g.a = fl ? prod_l[i] : -1;
    g.b = fr ? prod_r[i] : -1;
    g.lk = fl ? child_kind(ops[g.a]) : (P.flags & PLLGPU_OP_LEFT_TIP) ? CK_TIP : CK_INNER;
    g.rk = fr ? child_kind(ops[g.b]) : (P.flags & PLLGPU_OP_RIGHT_TIP) ? CK_TIP : CK_INNER;
    g.level = L;
    role[i] = 1;
    eff[i] = L;
    if (fl) role[g.a] = 2;
    if (fr) role[g.b] = 2;
    groups.push_back(g);
  }
}

static void to_fop(const DevOp &d, FOp &f)
{
  f.parent = d.parent;
  f.left = d.left;
  f.right = d.right;
  f.ltip = d.ltip;
  f.rtip = d.rtip;
  f.pscaler = d.pscaler;
  f.lscaler = d.lscaler;
  f.rscaler = d.rscaler;
  f.lmat = d.lmat;
  f.rmat = d.rmat;
}

template <int LK, int RK>
static void launch_fused_t(pllgpu_ctx *c, const FusePack &pack, unsigned ngroups, unsigned entries)
{
  const unsigned tiles = (entries + 63) / 64;
  const unsigned tpw = kDnaTilesPerWave;
  const unsigned nx = (tiles + 4 * tpw - 1) / (4 * tpw);
  // parents of this launch: entries x 128 B each. Beyond the 256 MB Infinity Cache nothing of them
  // survives until the next level reads it: stream them out as well
  unsigned stream_parent = ((size_t)ngroups * entries * 128u > c->stream_parent_bytes) ? 1u : 0u;
  // only the groups that are fed from tip codes are store traffic (kernels_common.h: xcd_block); the others keep the natural order
  const unsigned xcd = (c->xcd_order && LK != CK_INNER && LK != CK_FII && RK != CK_FII) ? 1u : 0u;
  hipLaunchKernelGGL((k_partials_dna_fused<LK, RK>), xcd_grid(nx, ngroups), dim3(256), 0, c->stream, pack, entries, c->gg.scale_mode, tpw, stream_parent,
                     nx, ngroups, xcd);
}

static void to_top(const DevOp &d, TOp &t)
{
  t.parent = d.parent;
  t.ltip = d.ltip;
  t.rtip = d.rtip;
  t.pscaler = d.pscaler;
  t.lmat = d.lmat;
  t.rmat = d.rmat;
}

template <int LK, int RK>
static void launch_cc_t(pllgpu_ctx *c, const CCPack &pack, unsigned ngroups, unsigned entries)
{
  const unsigned tiles = (entries + 63) / 64;
  const unsigned tpw = kDnaTilesPerWave;
  const unsigned nx = (tiles + 4 * tpw - 1) / (4 * tpw);
  unsigned stream_parent = ((size_t)ngroups * entries * 128u > c->stream_parent_bytes) ? 1u : 0u;
  hipLaunchKernelGGL((k_partials_dna_cc<LK, RK>), xcd_grid(nx, ngroups), dim3(256), 0, c->stream, pack, entries, c->gg.scale_mode, tpw, stream_parent,
                     nx, ngroups, c->xcd_order);
}

static int launch_cc16(pllgpu_ctx *c, const CC16Pack &pack, unsigned ngroups, unsigned entries)
{
  const unsigned tiles = (entries + 63) / 64;
  const unsigned nx = (tiles + 1) / 2; // a pair of waves per tile, two tiles per workgroup
  const unsigned stream_parent = ((size_t)ngroups * entries * 128u > c->stream_parent_bytes) ? 1u : 0u;
  hipLaunchKernelGGL(k_partials_dna_cc16, xcd_grid(nx, ngroups), dim3(256), 0, c->stream, pack, entries, c->gg.scale_mode, stream_parent, nx, ngroups,
                     c->xcd_order);
  return 0;
}

static int launch_cc(pllgpu_ctx *c, const CCPack &pack, unsigned ngroups, unsigned entries, int lk, int rk)
{
  if (lk == CK_INNER && rk == CK_FCC) launch_cc_t<CK_INNER, CK_FCC>(c, pack, ngroups, entries);
  else if (lk == CK_TIP && rk == CK_FCC) launch_cc_t<CK_TIP, CK_FCC>(c, pack, ngroups, entries);
  else if (lk == CK_FCC && rk == CK_FCC) launch_cc_t<CK_FCC, CK_FCC>(c, pack, ngroups, entries);
  else return fail(PLLGPU_EINVAL, "no cherry-cherry kernel for child kinds (%d, %d)", lk, rk);
  return 0;
}

static int launch_fused(pllgpu_ctx *c, const FusePack &pack, unsigned ngroups, unsigned entries, int lk, int rk)
{
#define FZ(A, B)                                          \
  if (lk == A && rk == B)                                 \
  {                                                       \
    launch_fused_t<A, B>(c, pack, ngroups, entries);      \
    return 0;                                             \
  }
  FZ(CK_INNER, CK_FTT) FZ(CK_INNER, CK_FTI) FZ(CK_INNER, CK_FII)
  FZ(CK_TIP, CK_FTT) FZ(CK_TIP, CK_FTI) FZ(CK_TIP, CK_FII)
  FZ(CK_FTT, CK_FTT) FZ(CK_FTT, CK_FTI) FZ(CK_FTT, CK_FII)
  FZ(CK_FTI, CK_FTI) FZ(CK_FTI, CK_FII) FZ(CK_FII, CK_FII)
#undef FZ
  return fail(PLLGPU_EINVAL, "no fused kernel for child kinds (%d, %d)", lk, rk);
}

// bytes one CLV update has to move: inner children, tip codes, the parent, the scaler vectors
static double op_traffic(const pllgpu_ctx *c, const pllgpu_op_t &o, bool read_left, bool read_right)
{
  const double span = (double)c->gg.S * c->gg.R * 8.0, sc = c->geo.per_rate_scalers ? 4.0 * c->gg.R : 4.0;
  double b = span + (o.parent_scaler >= 0 ? sc : 0.0);
  if (read_left) b += (o.flags & PLLGPU_OP_LEFT_TIP) ? 1.0 : span + (o.left_scaler >= 0 ? sc : 0.0);
  if (read_right) b += (o.flags & PLLGPU_OP_RIGHT_TIP) ? 1.0 : span + (o.right_scaler >= 0 ? sc : 0.0);
  return b * o.parent_entries;
}

// descriptor packs of the cherry-cherry groups [g0, g1) of one level, one per (memory-side kind, entries)
struct CCLaunch
{
  CCPack pack;
  unsigned n, entries;
  int lk;
};

static int build_cc_launches(pllgpu_ctx *c, const pllgpu_op_t *ops, const std::vector<FusedGroup> &groups, size_t g0, size_t g1,
                             std::vector<CCLaunch> &out)
{
  for (int lk = 0; lk <= CK_FCC; ++lk)
  {
    if (lk != CK_INNER && lk != CK_TIP && lk != CK_FCC) continue;
    CCLaunch cur;
    cur.n = 0;
    cur.entries = 0;
    cur.lk = lk;
    auto flush = [&]() {
      if (cur.n) out.push_back(cur);
      cur.n = 0;
    };
    for (size_t gi = g0; gi < g1; ++gi)
    {
      const FusedGroup &g = groups[gi];
      if (g.lk != CK_FCC && g.rk != CK_FCC) continue;
      if (g.absorbed) continue; // evaluated by a fifteen-op group (build_cc16_launches)
      const bool swap = g.lk == CK_FCC && g.rk != CK_FCC; // canonical order: the memory side on the left
      const int glk = swap ? g.rk : g.lk;
      if (glk != lk) continue;
      const pllgpu_op_t &P = ops[g.p];
      if (P.parent_entries == 0) continue;
      if (cur.n && P.parent_entries != cur.entries) flush();
      cur.entries = P.parent_entries;
      CCGroup &cg = cur.pack.g[cur.n];
      memset(&cg, 0, sizeof cg);
      DevOp d;
      auto side = [&](int a, int x, int y, FOp &fa, TOp &tx, TOp &ty) -> int {
        if (a < 0) return 0;
        if (int rc = resolve_op(c, ops[x], d)) return rc;
        to_top(d, tx);
        c->last_bytes += op_traffic(c, ops[x], true, true);
        if (int rc = resolve_op(c, ops[y], d)) return rc;
        to_top(d, ty);
        c->last_bytes += op_traffic(c, ops[y], true, true);
        if (int rc = resolve_op(c, ops[a], d)) return rc;
        to_fop(d, fa);
        c->last_bytes += op_traffic(c, ops[a], false, false);
        return 0;
      };
      if (int rc = side(g.a, g.aa, g.ab, cg.a, cg.aa, cg.ab)) return rc;
      if (int rc = side(g.b, g.ba, g.bb, cg.b, cg.ba, cg.bb)) return rc;
      if (int rc = resolve_op(c, P, d)) return rc;
      c->last_bytes += op_traffic(c, P, g.a < 0, g.b < 0);
      to_fop(d, cg.p);
      if (swap)
      {
        std::swap(cg.p.left, cg.p.right);
        std::swap(cg.p.ltip, cg.p.rtip);
        std::swap(cg.p.lscaler, cg.p.rscaler);
        std::swap(cg.p.lmat, cg.p.rmat);
        std::swap(cg.a, cg.b);
        std::swap(cg.aa, cg.ba);
        std::swap(cg.ab, cg.bb);
      }
      if (++cur.n == (unsigned)kMaxCCGroups) flush();
    }
    flush();
  }
  return 0;
}

// descriptor packs of the fifteen-op groups (kind CK_F8 on both sides: complete 16-tip subtrees), one per entry count
struct CC16Launch
{
  CC16Pack pack;
  unsigned n, entries;
};

// a complete (CK_FCC, CK_FCC) group as the kernels read it; counts its seven ops' bytes
static int fill_cc8(pllgpu_ctx *c, const pllgpu_op_t *ops, const FusedGroup &g, CCGroup &cg)
{
  memset(&cg, 0, sizeof cg);
  DevOp d;
  auto side = [&](int a, int x, int y, FOp &fa, TOp &tx, TOp &ty) -> int {
    if (int rc = resolve_op(c, ops[x], d)) return rc;
    to_top(d, tx);
    c->last_bytes += op_traffic(c, ops[x], true, true);
    if (int rc = resolve_op(c, ops[y], d)) return rc;
    to_top(d, ty);
    c->last_bytes += op_traffic(c, ops[y], true, true);
    if (int rc = resolve_op(c, ops[a], d)) return rc;
    to_fop(d, fa);
    c->last_bytes += op_traffic(c, ops[a], false, false);
    return 0;
  };
  if (int rc = side(g.a, g.aa, g.ab, cg.a, cg.aa, cg.ab)) return rc;
  if (int rc = side(g.b, g.ba, g.bb, cg.b, cg.ba, cg.bb)) return rc;
  if (int rc = resolve_op(c, ops[g.p], d)) return rc;
  c->last_bytes += op_traffic(c, ops[g.p], false, false);
  to_fop(d, cg.p);
  return 0;
}

static int build_cc16_launches(pllgpu_ctx *c, const pllgpu_op_t *ops, const std::vector<FusedGroup> &groups, std::vector<CC16Launch> &out)
{
  CC16Launch cur;
  cur.n = 0;
  cur.entries = 0;
  auto flush = [&]() {
    if (cur.n) out.push_back(cur);
    cur.n = 0;
  };
  for (const FusedGroup &g : groups)
  {
    if (g.lk != CK_F8) continue;
    const pllgpu_op_t &P = ops[g.p];
    if (P.parent_entries == 0) continue;
    if (cur.n && P.parent_entries != cur.entries) flush();
    cur.entries = P.parent_entries;
    CC16Group &cg = cur.pack.g[cur.n];
    memset(&cg, 0, sizeof cg);
    if (int rc = fill_cc8(c, ops, groups[g.ga], cg.a)) return rc;
    if (int rc = fill_cc8(c, ops, groups[g.gb], cg.b)) return rc;
    DevOp d;
    if (int rc = resolve_op(c, P, d)) return rc;
    c->last_bytes += op_traffic(c, P, false, false);
    to_fop(d, cg.p);
    if (++cur.n == (unsigned)kMaxCC16Groups) flush();
  }
  flush();
  return 0;
}
