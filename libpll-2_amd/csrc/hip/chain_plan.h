// chain_plan.h - part of the pllgpu.hip translation unit (included there, after fusion_plan.h): the
// partition of a dependency-only 4x4 op list into chains and stages, the cached
// plan with its descriptors, and the chain launches (kernels_dna.h: k_partials_dna_chain).
#pragma once
// ---- chain plans (DNA) ------------------------------------------------------------------------------
// For op lists whose only ordering constraints are producer -> consumer (every CLV and scaler written
// once, read by one later op at most, nothing overwritten that an earlier op touches): instead of one
// launch per dependency level, the ops are partitioned into CHAINS (kernels_dna.h:
// k_partials_dna_chain) - a path towards the root whose running CLV stays in registers - and the
// chains into STAGES: a chain runs in the first launch in which every CLV it reads from HBM exists.
//   S[i] = stage of op i = min over (a = child in registers, b = sibling) of
//          max( S[a]  (a leaf: 1),  b a leaf or the top of another chain: S[b] + 1,
//                                   b formed on the fly from two stored CLVs / tips: A[b] )
//   A[b] = 1 + max(S[children of b])
// computed bottom-up; the partition follows top-down from the ops nobody consumes. Complete 8-tip
// subtrees keep their own kernel (cherry-cherry groups, stage 1); their CLVs are leaves here.
struct ChainLaunchRec
{
  unsigned first_head, nchains; // one launch over heads [first_head, first_head + nchains)
  unsigned variant;             // which fetch groups the kernel issues: 0 tips only, 1 + a second tip (cherry siblings),
                                // 2 tips or a CLV, 3 everything
};

struct ChainPlan
{
  std::vector<pllgpu_op_t> key;
  unsigned long long epoch = 0;
  unsigned entries = 0;
  std::vector<CCLaunch> cc;            // stage 1, before the chains
  std::vector<CC16Launch> cc16;        // stage 1 as well: complete 16-tip subtrees (fifteen ops per group)
  std::vector<ChainLaunchRec> stages;
  std::vector<ChainHead> heads;
  std::vector<ChainStepLoad> loads;
  std::vector<ChainStepOp> sops;
  std::vector<unsigned> head_top_clv;  // per head: what its last step produces
  std::vector<int> head_top_scaler;
  std::vector<unsigned char> head_variant;
  bool in_kernarg = false;             // every stage fits a ChainPack
  size_t held_from = 0;                // stages[held_from ..) = the last stage when it may be held for the edge evaluation
  unsigned launches = 0;
  double bytes = 0.0;
};

static void drop_chain_plan(pllgpu_ctx *c)
{
  delete c->plan;
  c->plan = nullptr;
}

// one launch over the heads [first_head, first_head + nchains) of a plan
static void launch_chain_heads(pllgpu_ctx *c, const ChainPlan &pl, unsigned first_head, unsigned nchains, unsigned variant)
{
  const unsigned tiles = (pl.entries + 63) / 64;
  dim3 grid(tiles, nchains), block(256); // a workgroup = one 64-site tile, wave k = rate category k
  if (pl.in_kernarg)
  {
    ChainPack pack;
    memset(&pack, 0, sizeof pack);
    unsigned ns = 0;
    for (unsigned h = 0; h < nchains; ++h)
    {
      ChainHead hd = pl.heads[first_head + h];
      memcpy(&pack.loads[ns], &pl.loads[hd.first], (hd.nsteps + 1) * sizeof(ChainStepLoad)); // + the terminal step
      memcpy(&pack.ops[ns], &pl.sops[hd.first], (hd.nsteps + 1) * sizeof(ChainStepOp));
      hd.first = ns;
      ns += hd.nsteps + 1;
      pack.heads[h] = hd;
    }
#define CHAIN_PACK(SMV, C0, S1, C1) hipLaunchKernelGGL((k_partials_dna_chain_pack<SMV, C0, S1, C1>), grid, block, 0, c->stream, pack, pl.entries)
#define CHAIN_PACK_V(SMV)                              \
  switch (variant)                                     \
  {                                                    \
  case 0: CHAIN_PACK(SMV, false, false, false); break; \
  case 1: CHAIN_PACK(SMV, false, true, false); break;  \
  case 2: CHAIN_PACK(SMV, true, false, false); break;  \
  default: CHAIN_PACK(SMV, true, true, true); break;   \
  }
    if (c->gg.scale_mode == 2)
    {
      CHAIN_PACK_V(2)
    }
    else
    {
      CHAIN_PACK_V(1)
    }
#undef CHAIN_PACK_V
#undef CHAIN_PACK
  }
  else
  {
    const unsigned char *base = c->chain_dev.p;
    const size_t heads_bytes = pl.heads.size() * sizeof(ChainHead), loads_bytes = pl.loads.size() * sizeof(ChainStepLoad);
    const ChainHead *hp = reinterpret_cast<const ChainHead *>(base) + first_head;
    const ChainStepLoad *lp = reinterpret_cast<const ChainStepLoad *>(base + heads_bytes);
    const ChainStepOp *op = reinterpret_cast<const ChainStepOp *>(base + heads_bytes + loads_bytes);
#define CHAIN_MEM(SMV, C0, S1, C1) hipLaunchKernelGGL((k_partials_dna_chain<SMV, C0, S1, C1>), grid, block, 0, c->stream, hp, lp, op, pl.entries)
#define CHAIN_MEM_V(SMV)                              \
  switch (variant)                                    \
  {                                                   \
  case 0: CHAIN_MEM(SMV, false, false, false); break; \
  case 1: CHAIN_MEM(SMV, false, true, false); break;  \
  case 2: CHAIN_MEM(SMV, true, false, false); break;  \
  default: CHAIN_MEM(SMV, true, true, true); break;   \
  }
    if (c->gg.scale_mode == 2)
    {
      CHAIN_MEM_V(2)
    }
    else
    {
      CHAIN_MEM_V(1)
    }
#undef CHAIN_MEM_V
#undef CHAIN_MEM
  }
}

// the whole plan, or - hold = true - everything but its last stage, which stays with the context until
// the next call shows whether it is the evaluation of the edge those chains end in
static int launch_chain_plan(pllgpu_ctx *c, const ChainPlan &pl, bool hold)
{
  for (const CC16Launch &l : pl.cc16)
    if (int rc = launch_cc16(c, l.pack, l.n, l.entries)) return rc;
  for (const CCLaunch &l : pl.cc)
    if (int rc = launch_cc(c, l.pack, l.n, l.entries, l.lk, CK_FCC)) return rc;
  const size_t upto = hold ? pl.held_from : pl.stages.size();
  for (size_t i = 0; i < upto; ++i) launch_chain_heads(c, pl, pl.stages[i].first_head, pl.stages[i].nchains, pl.stages[i].variant);
  c->chain_held = hold && upto < pl.stages.size();
  c->last_launches = pl.launches - (unsigned)(pl.stages.size() - upto);
  c->last_bytes = pl.bytes;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PLLGPU_ERUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

static int launch_held_chains(pllgpu_ctx *c)
{
  if (!c->chain_held || !c->plan) return 0;
  c->chain_held = false;
  const ChainPlan &pl = *c->plan;
  for (size_t i = pl.held_from; i < pl.stages.size(); ++i)
  {
    launch_chain_heads(c, pl, pl.stages[i].first_head, pl.stages[i].nchains, pl.stages[i].variant);
    ++c->last_launches;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PLLGPU_ERUNTIME, "kernel launch failed: %s", hipGetErrorString(e));
  return 0;
}

struct PChain
{
  std::vector<unsigned> ops; // top first
  unsigned stage;
};

struct ChainPartition
{
  std::vector<int> pr_of[2];  // producer op of the left / right child, or -1
  std::vector<int> role;      // != 0: member of a cherry-cherry group (plan_fusion)
  std::vector<FusedGroup> groups;
  std::vector<unsigned> S;    // launch stage of every chain op
  std::vector<unsigned char> acc_side, absorb, form; // form: 0 top of a chain, 1 step below the next op of its chain, 2 formed on the fly as a sibling
  std::vector<int> chain_of;
  std::vector<PChain> chains;
};

// host logic only (no device state): does the list qualify, and how is it partitioned
static bool partition_chains(const pllgpu_op_t *ops, unsigned count, unsigned nodes, unsigned nsb, unsigned entries, bool fuse_cc, bool fuse_cc16,
                             ChainPartition &P)
{
  // ---- is the list dependency-only?
  std::vector<int> producer(nodes, -1), sc_writer(nsb, -1), consumers(count, 0);
  std::vector<int> (&pr_of)[2] = P.pr_of;
  pr_of[0].assign(count, -1);
  pr_of[1].assign(count, -1);
  for (unsigned i = 0; i < count; ++i)
  {
    const pllgpu_op_t &o = ops[i];
    if ((o.flags & PLLGPU_OP_GATHER) || o.parent_entries != entries || o.war_level >= 0 || o.left_clv == o.right_clv) return false;
    if ((o.flags & PLLGPU_OP_RIGHT_TIP) && !(o.flags & PLLGPU_OP_LEFT_TIP)) return false; // the level path reports it
    if (o.parent_clv >= nodes || o.left_clv >= nodes || o.right_clv >= nodes || producer[o.parent_clv] >= 0) return false;
    if (o.parent_scaler >= (int)nsb || o.left_scaler >= (int)nsb || o.right_scaler >= (int)nsb) return false;
    const unsigned kid[2] = {o.left_clv, o.right_clv};
    const int ksc[2] = {o.left_scaler, o.right_scaler};
    const bool tip[2] = {(o.flags & PLLGPU_OP_LEFT_TIP) != 0, (o.flags & PLLGPU_OP_RIGHT_TIP) != 0};
    for (int sd = 0; sd < 2; ++sd)
    {
      if (tip[sd]) continue;
      const int pr = producer[kid[sd]];
      if (pr >= 0)
      {
        if (++consumers[pr] > 1 || ops[pr].parent_scaler != ksc[sd]) return false;
        pr_of[sd][i] = pr;
      }
      else if (ksc[sd] >= 0 && sc_writer[ksc[sd]] >= 0)
        return false; // a stored CLV paired with a scaler this list rewrites
    }
    producer[o.parent_clv] = (int)i;
    if (o.parent_scaler >= 0)
    {
      if (sc_writer[o.parent_scaler] >= 0) return false;
      sc_writer[o.parent_scaler] = (int)i;
    }
  }
  // a tip child whose codes were replaced by a dense CLV arrives as an inner child: nothing to do here
  std::vector<int> &role = P.role;
  std::vector<FusedGroup> &groups = P.groups;
  plan_fusion(true, fuse_cc, nodes, ops, count, role, groups, true, fuse_cc16); // cherry-cherry groups (and groups of two of them) only
  // ---- stages, bottom-up
  std::vector<unsigned> &S = P.S;
  std::vector<unsigned char> &acc_side = P.acc_side, &absorb = P.absorb;
  S.assign(count, 0);
  acc_side.assign(count, 0);
  absorb.assign(count, 0);
  auto leaf_ready = [&](unsigned i, int sd) -> unsigned { // stage after which a non-chain child exists in HBM
    const int pr = pr_of[sd][i];
    return (pr >= 0 && role[pr] != 0) ? 1u : 0u; // cherry-cherry groups run in stage 1
  };
  auto is_chain_op = [&](unsigned i, int sd) { return pr_of[sd][i] >= 0 && role[pr_of[sd][i]] == 0; };
  auto ready_in_hbm = [&](unsigned i, int sd) -> unsigned { return is_chain_op(i, sd) ? S[pr_of[sd][i]] : leaf_ready(i, sd); };
  for (unsigned i = 0; i < count; ++i)
  {
    if (role[i]) continue;
    unsigned best = ~0u, best_cost = ~0u;
    for (int a = 0; a < 2; ++a)
    {
      const int b = 1 - a;
      const bool tip_a = ops[i].flags & (a ? PLLGPU_OP_RIGHT_TIP : PLLGPU_OP_LEFT_TIP);
      const bool tip_b = ops[i].flags & (b ? PLLGPU_OP_RIGHT_TIP : PLLGPU_OP_LEFT_TIP);
      unsigned req_a, cost = 0;
      if (is_chain_op(i, a))
        req_a = S[pr_of[a][i]];
      else
      {
        req_a = leaf_ready(i, a) + 1;
        cost += tip_a ? 1u : 132u;
      }
      unsigned req_b;
      bool ab = false;
      if (is_chain_op(i, b))
      {
        const unsigned q = (unsigned)pr_of[b][i];
        const unsigned A = 1 + std::max(ready_in_hbm(q, 0), ready_in_hbm(q, 1));
        if (A <= S[q])
        {
          ab = true;
          req_b = A;
        }
        else
        {
          req_b = S[q] + 1;
          cost += 132u;
        }
      }
      else
      {
        req_b = leaf_ready(i, b) + 1;
        cost += tip_b ? 1u : 132u;
      }
      const unsigned st = std::max(req_a, req_b);
      if (st < best || (st == best && cost < best_cost))
      {
        best = st;
        best_cost = cost;
        acc_side[i] = (unsigned char)a;
        absorb[i] = ab ? 1 : 0;
      }
    }
    S[i] = best;
  }
  // ---- partition, top-down: form 0 = top of a chain (default), 1 = step below the next op of its chain, 2 = formed
  // on the fly as a sibling
  std::vector<unsigned char> &form = P.form;
  std::vector<int> &chain_of = P.chain_of;
  std::vector<PChain> &chains = P.chains;
  form.assign(count, 0);
  chain_of.assign(count, -1);
  chains.clear();
  for (unsigned ii = count; ii-- > 0;)
  {
    const unsigned i = ii;
    if (role[i] || form[i] == 2) continue;
    if (form[i] == 0)
    {
      chain_of[i] = (int)chains.size();
      chains.push_back(PChain{{}, S[i]});
    }
    chains[chain_of[i]].ops.push_back(i);
    const int a = acc_side[i], b = 1 - a;
    if (is_chain_op(i, a))
    {
      form[pr_of[a][i]] = 1;
      chain_of[pr_of[a][i]] = chain_of[i];
    }
    if (is_chain_op(i, b) && absorb[i]) form[pr_of[b][i]] = 2;
  }
  return true;
}

extern "C" int pllgpu_debug_chain_plan(const pllgpu_op_t *ops, unsigned count, unsigned nodes, unsigned scale_buffers, int fuse_cc,
                                       unsigned *stage, int *chain, unsigned char *form)
{
  if (!ops || count == 0) return 0;
  ChainPartition P;
  // fuse_cc: bit 0 = seven-op groups, bit 1 = fifteen-op groups over two of them
  if (!partition_chains(ops, count, nodes, scale_buffers, ops[0].parent_entries, (fuse_cc & 1) != 0, (fuse_cc & 2) != 0, P)) return 0;
  unsigned stages = 0;
  for (unsigned i = 0; i < count; ++i)
  {
    const bool member = P.role[i] != 0;
    unsigned st = member ? 1u : P.S[i];
    if (!member && P.chain_of[i] >= 0) st = P.chains[P.chain_of[i]].stage; // a chain runs where its top runs
    if (!member && P.form[i] == 2)
      for (unsigned j = i + 1; j < count; ++j) // the step that forms it on the fly
        if (P.role[j] == 0 && P.absorb[j] && P.pr_of[1 - P.acc_side[j]][j] == (int)i) st = P.chains[P.chain_of[j]].stage;
    if (stage) stage[i] = st;
    if (chain) chain[i] = member ? -1 : (P.form[i] == 2 ? -2 : P.chain_of[i]);
    if (form) form[i] = member ? 3 : P.form[i];
    stages = std::max(stages, st);
  }
  return (int)stages;
}

// Fifteen-op groups pay where the launch is long: same box, alternating (tools/round4_calls/r4_cc16.sh) - 64 taxa x 100k
// sites 0.160-0.162 ms per step with either form (the group launch gains what the shorter chain tail saves: nothing
// left over), 128 taxa x 100k sites +7 %, 64 taxa x 200k / 400k sites +8 / +12 %. By size unless PLL_AMD_FUSE_CC16 says
// 0 / 1: entries x ops of the list, the crossover between the first and the other cases.
static bool use_cc16(const pllgpu_ctx *c, unsigned entries, unsigned count)
{
  if (c->fuse_cc16 >= 0) return c->fuse_cc16 != 0;
  return (size_t)entries * count >= (size_t)9000000;
}

// returns 0 and sets used = true when the list was planned and launched as chains
static int try_chain_plan(pllgpu_ctx *c, const pllgpu_op_t *ops, unsigned count, bool &used)
{
  used = false;
  if (!c->chains || !c->fuse || count < 4 || c->any_aos) return 0; // up to three ops the level scheduler + tail kernel is as fast (tools/path_latency.py)
  if (c->plan && c->plan->epoch == c->alloc_epoch && c->plan->key.size() == count &&
      memcmp(c->plan->key.data(), ops, count * sizeof(pllgpu_op_t)) == 0)
  {
    used = true;
    ++c->plan_replays;
    return launch_chain_plan(c, *c->plan, c->defer_tail);
  }
  const unsigned entries = ops[0].parent_entries;
  if (entries == 0 || aos_entries(c, entries) || (size_t)entries * 128u >= ((size_t)1 << 31)) return 0; // 32-bit buffer offsets
  const unsigned nodes = c->geo.nodes, nsb = c->geo.scale_buffers;
  ChainPartition part;
  if (!partition_chains(ops, count, nodes, nsb, entries, c->fuse_cc, use_cc16(c, entries, count), part)) return 0;
  ChainPlan *pl = new ChainPlan();
  std::vector<int> (&pr_of)[2] = part.pr_of;
  std::vector<int> &role = part.role;
  std::vector<FusedGroup> &groups = part.groups;
  std::vector<unsigned> &S = part.S;
  std::vector<unsigned char> &acc_side = part.acc_side, &absorb = part.absorb;
  std::vector<PChain> &chains = part.chains;
  auto is_chain_op = [&](unsigned i, int sd) { return pr_of[sd][i] >= 0 && role[pr_of[sd][i]] == 0; };
  (void)S;
  // ---- descriptors: resolve every op once, in list order (a producer's buffers exist before its consumer looks)
  std::vector<DevOp> dev(count);
  c->last_bytes = 0.0;
  for (unsigned i = 0; i < count; ++i)
    if (role[i] == 0)
      if (int rc = resolve_op(c, ops[i], dev[i]))
      {
        delete pl;
        return rc;
      }
  {
    sort_groups_by_level(groups); // (ga / gb of the fifteen-op groups are indices into this list: they move with it)
    if (int rc = build_cc_launches(c, ops, groups, 0, groups.size(), pl->cc))
    {
      delete pl;
      return rc;
    }
    if (int rc = build_cc16_launches(c, ops, groups, pl->cc16))
    {
      delete pl;
      return rc;
    }
  }
  const unsigned clv_bytes = (unsigned)(clv_elems(c, entries) * sizeof(double));
  const unsigned sc_bytes = entries * (c->gg.scale_mode == 2 ? 16u : 4u);
  auto leaf_of = [&](unsigned i, int sd, bool &tip, ChainLeafBytes &lb) -> ChainLeaf {
    ChainLeaf l;
    tip = ops[i].flags & (sd ? PLLGPU_OP_RIGHT_TIP : PLLGPU_OP_LEFT_TIP);
    lb.clv = lb.aux = lb.pad = 0u;
    if (tip)
    {
      l.data = sd ? dev[i].rtip : dev[i].ltip;
      l.scaler = nullptr;
      lb.aux = (entries + 3u) & ~3u; // the codes are read four at a time (hipMalloc blocks are padded well beyond that)
    }
    else
    {
      l.data = sd ? dev[i].right : dev[i].left;
      l.scaler = sd ? dev[i].rscaler : dev[i].lscaler;
      lb.clv = clv_bytes;
      lb.aux = l.scaler ? sc_bytes : 0u;
    }
    return l;
  };
  unsigned max_stage = 0;
  for (const PChain &ch : chains) max_stage = std::max(max_stage, ch.stage);
  pl->entries = entries;
  pl->in_kernarg = true;
  pl->held_from = (size_t)-1;
  // what a chain's steps have to fetch decides the kernel variant it runs under (kernels_dna.h)
  auto variant_of = [&](const PChain &ch) -> unsigned {
    bool c0 = false, s1 = false, c1 = false;
    for (unsigned i : ch.ops)
    {
      const int b = 1 - acc_side[i];
      if (is_chain_op(i, b) && absorb[i])
      {
        const unsigned q = (unsigned)pr_of[b][i];
        const bool t0 = ops[q].flags & PLLGPU_OP_LEFT_TIP, t1 = ops[q].flags & PLLGPU_OP_RIGHT_TIP;
        s1 = true;
        if (!t0) c0 = true;
        if (!t1) c1 = true;
      }
      else if (!(ops[i].flags & (b ? PLLGPU_OP_RIGHT_TIP : PLLGPU_OP_LEFT_TIP)))
        c0 = true;
    }
    if (c1 || (c0 && s1)) return 3u;
    return c0 ? 2u : s1 ? 1u : 0u;
  };
  std::vector<unsigned> chain_variant(chains.size());
  for (unsigned k = 0; k < chains.size(); ++k) chain_variant[k] = variant_of(chains[k]);
  for (unsigned st = 1; st <= max_stage; ++st)
   for (unsigned variant = 0; variant < 4; ++variant)
  {
    std::vector<unsigned> ids;
    for (unsigned k = 0; k < chains.size(); ++k)
      if (chains[k].stage == st && chain_variant[k] == variant) ids.push_back(k);
    if (ids.empty()) continue;
    // the longest chains first: their workgroups are dispatched first
    std::stable_sort(ids.begin(), ids.end(), [&](unsigned x, unsigned y) { return chains[x].ops.size() > chains[y].ops.size(); });
    ChainLaunchRec rec;
    rec.first_head = (unsigned)pl->heads.size();
    rec.nchains = (unsigned)ids.size();
    rec.variant = variant;
    const bool stream_tops = (size_t)ids.size() * entries * 128u > c->stream_parent_bytes;
    unsigned stage_steps = 0;
    for (unsigned k : ids)
    {
      const PChain &ch = chains[k];
      ChainHead hd;
      memset(&hd, 0, sizeof hd);
      hd.first = (unsigned)pl->loads.size();
      hd.nsteps = (unsigned)ch.ops.size();
      stage_steps += hd.nsteps;
      for (size_t t = ch.ops.size(); t-- > 0;) // bottom first
      {
        const unsigned i = ch.ops[t];
        const int a = acc_side[i], b = 1 - a;
        const bool bottom = t + 1 == ch.ops.size(), top = t == 0;
        if (bottom)
        {
          bool tip;
          hd.acc0 = leaf_of(i, a, tip, hd.bacc);
          hd.acc_tip = tip ? 1u : 0u;
        }
        ChainStepLoad ld;
        ChainStepOp so;
        memset(&ld, 0, sizeof ld);
        memset(&so, 0, sizeof so);
        so.parent = dev[i].parent;
        so.pscaler = dev[i].pscaler;
        so.mat_acc = a ? dev[i].rmat : dev[i].lmat;
        so.mat_sib = b ? dev[i].rmat : dev[i].lmat;
        so.p_bytes = clv_bytes;
        so.psc_bytes = so.pscaler ? sc_bytes : 0u;
        // the tops of the last stage are the ends of the edge evaluated next (from registers, chain tail):
        // nobody reads them back soon either
        if (!top || stream_tops || (c->defer_tail && st == max_stage)) ld.flags |= kChStream;
        bool read_sib = true;
        if (is_chain_op(i, b) && absorb[i])
        {
          const unsigned q = (unsigned)pr_of[b][i];
          bool t0, t1;
          ld.s0 = leaf_of(q, 0, t0, ld.b0); // a tip-inner op carries its tip on the left
          ld.s1 = leaf_of(q, 1, t1, ld.b1);
          ld.flags |= (t0 && t1) ? CS_OTT : t0 ? CS_OTC : CS_OCC;
          so.bparent = dev[q].parent;
          so.bpscaler = dev[q].pscaler;
          so.bmat0 = dev[q].lmat;
          so.bmat1 = dev[q].rmat;
          so.b_bytes = clv_bytes;
          so.bsc_bytes = so.bpscaler ? sc_bytes : 0u;
          pl->bytes += op_traffic(c, ops[q], true, true);
          read_sib = false;
        }
        else
        {
          bool t0;
          ld.s0 = leaf_of(i, b, t0, ld.b0);
          ld.flags |= t0 ? CS_T : CS_C;
        }
        pl->bytes += op_traffic(c, ops[i], a == 0 ? bottom : read_sib, a == 0 ? read_sib : bottom);
        pl->loads.push_back(ld);
        pl->sops.push_back(so);
      }
      {
        // the terminal step: what the last trip "prefetches" - every size 0
        ChainStepLoad ld;
        ChainStepOp so;
        memset(&ld, 0, sizeof ld);
        memset(&so, 0, sizeof so);
        ld.flags = CS_END;
        pl->loads.push_back(ld);
        pl->sops.push_back(so);
        ++stage_steps;
      }
      pl->heads.push_back(hd);
      pl->head_top_clv.push_back(ops[ch.ops[0]].parent_clv);
      pl->head_top_scaler.push_back(ops[ch.ops[0]].parent_scaler);
      pl->head_variant.push_back((unsigned char)variant);
    }
    if (st == max_stage && pl->held_from == (size_t)-1) pl->held_from = pl->stages.size();
    if (rec.nchains > (unsigned)kChainPackHeads || stage_steps + 1 > (unsigned)kChainPackSteps) pl->in_kernarg = false; // + 1: a chain of no steps in the tail
    pl->stages.push_back(rec);
  }
  {
    // the last stage may wait for the edge evaluation if it is at most the two ends of an edge and the
    // tail kernel can reproduce k_edge_dna's summation order (one tile per wave there: <= 4096 tiles)
    if (pl->held_from == (size_t)-1) pl->held_from = pl->stages.size();
    unsigned held_chains = 0;
    for (size_t i = pl->held_from; i < pl->stages.size(); ++i) held_chains += pl->stages[i].nchains;
    if (held_chains > 2 || (c->geo.sites + 63) / 64 > 4096u) pl->held_from = pl->stages.size();
    // the tail kernel takes BOTH ends in one descriptor pack (an end that is not held counts one
    // terminal step); every stage record was sized against the pack on its own, two records of the last
    // stage (different fetch variants) together may not fit: such a plan keeps its descriptors in memory
    unsigned tail_steps = 0, tail_heads = 0;
    for (size_t i = pl->held_from; i < pl->stages.size(); ++i)
      for (unsigned h = pl->stages[i].first_head; h < pl->stages[i].first_head + pl->stages[i].nchains; ++h, ++tail_heads)
        tail_steps += pl->heads[h].nsteps + 1;
    if (tail_heads && tail_steps + (2u - std::min(tail_heads, 2u)) > (unsigned)kChainPackSteps) pl->in_kernarg = false;
  }
  pl->bytes += c->last_bytes; // the cherry-cherry groups (build_cc_launches counted them)
  pl->launches = (unsigned)(pl->cc.size() + pl->cc16.size() + pl->stages.size());
  if (!pl->in_kernarg)
  {
    const size_t hb = pl->heads.size() * sizeof(ChainHead), lb = pl->loads.size() * sizeof(ChainStepLoad),
                 ob = pl->sops.size() * sizeof(ChainStepOp);
    if (int rc = c->chain_dev.ensure(hb + lb + ob))
    {
      delete pl;
      return rc;
    }
    // pageable sources are staged before hipMemcpyAsync returns; the stream orders the copies behind the
    // kernels of the previous plan that still read the old descriptors
    // (through the context's pinned block: a tree search plans a new list after every move)
    hipError_t e = copy_up(c, c->chain_dev.p, pl->heads.data(), hb);
    if (e == hipSuccess) e = copy_up(c, c->chain_dev.p + hb, pl->loads.data(), lb);
    if (e == hipSuccess) e = copy_up(c, c->chain_dev.p + hb + lb, pl->sops.data(), ob);
    if (e != hipSuccess)
    {
      delete pl;
      return fail(PLLGPU_ERUNTIME, "descriptor upload failed: %s", hipGetErrorString(e));
    }
  }
  pl->key.assign(ops, ops + count);
  pl->epoch = c->alloc_epoch;
  drop_chain_plan(c);
  c->plan = pl;
  used = true;
  return launch_chain_plan(c, *pl, c->defer_tail);
}

