// subtree_plan.h - part of the pllgpu.hip translation unit (included there, after fusion_plan.h): site
// repeats, which ops of a level-scheduled 4x4 list are evaluated straight from the tip codes
// (kernels_dna.h: k_partials_dna_sub) and the one launch that does it.
#pragma once

// An op qualifies when
//   * it gathers (site repeats) and no group claimed it,
//   * each child is a tip given by codes, or the parent of an EARLIER qualifying op of this list whose
//     scaler is the one this op reads - so the whole subtree is tips and at most three ops deep,
//   * its child maps are indexed by its own entries (maps built on the device, or an uncompressed
//     parent whose entries are the sites),
//   * nothing earlier in the list reads or writes its outputs (war_level): the launch runs before every
//     level.
// Returns the ops claimed (role 4) through `role`; descriptors go to the context's device array, which is
// re-used as is when the same descriptors come again.
struct SubOpRec // one qualifying op while the list is planned
{
  DevOp d;
  int lchild, rchild; // SubOpRec of the child, or -1: a tip
  unsigned depth;
  bool has_scaler;
};

// position `pos` of `item` <- the subtree below rec r
static void flatten_subtree(const std::vector<SubOpRec> &recs, int r, unsigned pos, SubItem &item)
{
  const SubOpRec &x = recs[r];
  item.node_mask |= 1u << pos;
  if (x.has_scaler) item.scaler_mask |= 1u << pos;
  SubNode &nd = item.node[pos];
  nd.lmat = x.d.lmat;
  nd.rmat = x.d.rmat;
  nd.lent = x.d.lsid;
  nd.rent = x.d.rsid;
  const unsigned lp = 2 * pos + 1, rp = 2 * pos + 2;
  if (x.lchild >= 0)
    flatten_subtree(recs, x.lchild, lp, item);
  else
  {
    item.tip_mask |= 1u << lp;
    item.tip[lp - 1] = x.d.ltip;
  }
  if (x.rchild >= 0)
    flatten_subtree(recs, x.rchild, rp, item);
  else
  {
    item.tip_mask |= 1u << rp;
    item.tip[rp - 1] = x.d.rtip;
  }
}

static int plan_subtrees(pllgpu_ctx *c, const pllgpu_op_t *ops, unsigned count, std::vector<int> &role, unsigned &nsub, unsigned &max_entries)
{
  nsub = 0;
  max_entries = 0;
  if (!c->subtrees || !c->dna_fast) return 0;
  bool any = false;
  for (unsigned i = 0; i < count && !any; ++i) any = (ops[i].flags & PLLGPU_OP_GATHER) != 0;
  if (!any) return 0;
  const unsigned nodes = c->geo.nodes;
  std::vector<int> producer(nodes, -1), sub_of(count, -1);
  std::vector<SubOpRec> recs;
  std::vector<SubItem> &sub = c->sub_build;
  sub.clear();
  double bytes = 0.0;
  for (unsigned i = 0; i < count; ++i)
  {
    const pllgpu_op_t &o = ops[i];
    const int pl = (o.flags & PLLGPU_OP_LEFT_TIP) ? -1 : producer[o.left_clv];
    const int pr = (o.flags & PLLGPU_OP_RIGHT_TIP) ? -1 : producer[o.right_clv];
    producer[o.parent_clv] = (int)i;
    if (role[i] || !(o.flags & PLLGPU_OP_GATHER) || o.war_level >= 0 || o.parent_entries == 0) continue;
    if ((o.flags & PLLGPU_OP_RIGHT_TIP) && !(o.flags & PLLGPU_OP_LEFT_TIP)) continue; // the level path reports it
    unsigned d = 0;
    bool ok = true;
    auto side = [&](bool tip, int p, int want_scaler) {
      if (tip) return;
      if (p < 0 || sub_of[p] < 0 || ops[p].parent_scaler != want_scaler)
        ok = false;
      else
        d = std::max(d, recs[sub_of[p]].depth);
    };
    side(o.flags & PLLGPU_OP_LEFT_TIP, pl, o.left_scaler);
    side(o.flags & PLLGPU_OP_RIGHT_TIP, pr, o.right_scaler);
    if (!ok || d + 1 > 3 || (pl >= 0 && pl == pr)) continue;
    SubOpRec rec;
    if (int rc = resolve_op(c, o, rec.d)) return rc;
    if (!(rec.d.layout & kDirectMaps) && rec.d.id_site) continue; // maps by site behind an entry -> site map: level path
    rec.lchild = pl >= 0 ? sub_of[pl] : -1;
    rec.rchild = pr >= 0 ? sub_of[pr] : -1;
    rec.depth = d + 1;
    rec.has_scaler = o.parent_scaler >= 0;
    sub_of[i] = (int)recs.size();
    recs.push_back(rec);
    SubItem item;
    memset(&item, 0, sizeof item);
    item.parent = rec.d.parent;
    item.pscaler = rec.d.pscaler;
    item.entries = o.parent_entries;
    item.flags = (rec.d.layout & kAosParent) ? kSubAos : 0u;
    flatten_subtree(recs, sub_of[i], 0, item);
    sub.push_back(item);
    role[i] = 4;
    max_entries = std::max(max_entries, o.parent_entries);
    // as grouped: the op's CLV and scaler go out, what comes in is one code per tip of the subtree and the maps
    bytes += (double)o.parent_entries * ((double)c->gg.S * c->gg.R * 8.0 + (o.parent_scaler >= 0 ? (c->geo.per_rate_scalers ? 16.0 : 4.0) : 0.0) +
                                         (double)(1u << (d + 1)) + 8.0);
  }
  nsub = (unsigned)sub.size();
  if (nsub)
  {
    c->last_bytes += bytes;
    // where k_sub_pack leaves each op's packed tip codes
    size_t total = 0;
    for (const SubItem &it : sub) total += ((size_t)it.entries + 63u) & ~(size_t)63u;
    if (int rc = c->sub_packed.ensure(total)) return rc;
    size_t off = 0;
    for (SubItem &it : sub)
    {
      it.packed = c->sub_packed.p + off;
      off += ((size_t)it.entries + 63u) & ~(size_t)63u;
    }
  }
  return 0;
}

// the descriptor array on the device holds `items` (uploaded unless it already does)
static int upload_subtrees(pllgpu_ctx *c, const std::vector<SubItem> &items)
{
  const size_t nbytes = items.size() * sizeof(SubItem);
  const unsigned long long epoch = c->alloc_epoch;
  if (c->sub_epoch == epoch && c->sub_cache.size() == items.size() && memcmp(c->sub_cache.data(), items.data(), nbytes) == 0) return 0;
  if (int rc = c->sub_dev.ensure(nbytes)) return rc;
  // pageable source: staged before hipMemcpyAsync returns; ordered behind the kernels that read the old array
  HIP_TRY(copy_up(c, c->sub_dev.p, items.data(), nbytes));
  c->sub_pack_valid = false; // other descriptors: their packed codes have to be formed
  c->sub_cache = items;
  c->sub_epoch = c->alloc_epoch;
  return 0;
}

// a workgroup = one 64-entry tile of one op, wave k = rate category k (kernels_dna.h); up to 127 ops per launch
static unsigned launch_subtrees(pllgpu_ctx *c, unsigned nsub)
{
  const SubItem *items = reinterpret_cast<const SubItem *>(c->sub_dev.p);
  unsigned launches = 0;
  // formed anew when descriptors, tip data or host-built maps changed; when only the class kernels have rewritten maps
  // since, the launch looks at what they reported (k_sub_pack)
  const bool stale = !c->sub_pack_valid || c->sub_pack_foreign != c->maps_foreign || c->sub_pack_tips != c->tips_epoch;
  const bool rewritten = c->sub_pack_maps != c->maps_version;
  const unsigned *changed = stale || c->sub_pack_always ? nullptr : c->rep_changed.p;
  for (unsigned first = 0; first < nsub; first += (unsigned)kSubItemsPerLaunch, ++launches)
  {
    const unsigned n = std::min(nsub - first, (unsigned)kSubItemsPerLaunch);
    SubTiles tiles;
    unsigned t = 0;
    for (unsigned i = 0; i < n; ++i)
    {
      tiles.first[i] = t;
      t += (c->sub_cache[first + i].entries + 63u) / 64u;
    }
    for (unsigned i = n; i <= (unsigned)kSubItemsPerLaunch; ++i) tiles.first[i] = t;
    dim3 grid(t), block(256);
    // the entries' tip codes, packed: once per set of class maps, tip data and descriptors (kernels_dna.h: k_sub_pack)
    if (stale || rewritten) hipLaunchKernelGGL(k_sub_pack, dim3((t + 3u) / 4u), block, 0, c->stream, items + first, tiles, n, t, changed, c->sub_pack_since);
    if (c->gg.scale_mode == 2)
      hipLaunchKernelGGL(k_partials_dna_sub<2>, grid, block, 0, c->stream, items + first, tiles, n);
    else
      hipLaunchKernelGGL(k_partials_dna_sub<1>, grid, block, 0, c->stream, items + first, tiles, n);
  }
  c->sub_pack_valid = true;
  c->sub_pack_maps = c->maps_version;
  c->sub_pack_foreign = c->maps_foreign;
  c->sub_pack_tips = c->tips_epoch;
  c->sub_pack_since = c->rep_seq + 1u; // the next class-map call's sequence number
  return launches;
}
