// engine_topo.cpp -- topology preprocessing of a registered replica (immutable, shared by every quadrature point that uses it)
#include "engine.h"

namespace scema_eng {

// -------------------------------------------------------------------------------------------
// topology preprocessing
// -------------------------------------------------------------------------------------------
int build_topo(scema_md_engine *e, const scema_md_system *s, Topo &t) {
  const int n = s->natoms;
  if (n <= 0 || s->ntypes <= 0) return fail(e, SCEMA_MD_ERR_ARG, "natoms/ntypes out of range");
  if (n > MD_JMASK) return fail(e, SCEMA_MD_ERR_ARG, "too many atoms for the 27-bit neighbour index");
  t.natoms = n;
  // Lennard-Jones classes: atom types with identical eps/sigma rows are one class on the device (force-field
  // generators hand out an atom type per atom name; OPLS-AA has a dozen distinct Lennard-Jones sites).  Masses stay
  // per atom, charges are per atom anyway; the kernels index pair tables of ncls x ncls entries.
  const int nty = s->ntypes;
  std::vector<int> cls(nty, -1), rep;
  for (int u = 0; u < nty; u++) {
    for (size_t c = 0; c < rep.size() && cls[u] < 0; c++) {
      const int v = rep[c];
      bool same = true;
      for (int w = 0; w < nty && same; w++)
        same = s->eps[(size_t)u * nty + w] == s->eps[(size_t)v * nty + w] && s->sigma[(size_t)u * nty + w] == s->sigma[(size_t)v * nty + w] &&
               s->eps[(size_t)w * nty + u] == s->eps[(size_t)w * nty + v] && s->sigma[(size_t)w * nty + u] == s->sigma[(size_t)w * nty + v];
      if (same) cls[u] = (int)c;
    }
    if (cls[u] < 0) { cls[u] = (int)rep.size(); rep.push_back(u); }
  }
  const int ncls = (int)rep.size();
  if (ncls > MD_MAXTYPES) return fail(e, SCEMA_MD_ERR_ARG, "%d distinct Lennard-Jones types (of %d atom types): at most %d are supported", ncls, nty, MD_MAXTYPES);
  t.ntypes = ncls;
  t.type.resize(n);
  t.q.assign(s->charge, s->charge + n);
  t.mass_atom.resize(n);
  for (int i = 0; i < n; i++) {
    if (s->type[i] < 0 || s->type[i] >= nty) return fail(e, SCEMA_MD_ERR_ARG, "atom type out of range");
    t.type[i] = cls[s->type[i]];
    t.mass_atom[i] = s->mass[s->type[i]];
    t.qsqsum += t.q[i] * t.q[i];
    t.qsum += t.q[i];
  }
  const int nt2 = ncls * ncls;
  t.lj.resize(4 * nt2);
  for (int a = 0; a < ncls; a++)
    for (int b = 0; b < ncls; b++) {
      const size_t src = (size_t)rep[a] * nty + rep[b];
      const int k = a * ncls + b;
      const double s6 = std::pow(s->sigma[src], 6.0), s12 = s6 * s6;
      t.lj[k] = 48.0 * s->eps[src] * s12;
      t.lj[nt2 + k] = 24.0 * s->eps[src] * s6;
      t.lj[2 * nt2 + k] = 4.0 * s->eps[src] * s12;
      t.lj[3 * nt2 + k] = 4.0 * s->eps[src] * s6;
    }
  // ---- bond graph -> 1-2 / 1-3 / 1-4 partners (lowest level wins) ----
  std::vector<std::vector<int>> adj(n);
  double max_r0 = 0.0;
  for (int b = 0; b < s->nbonds; b++) {
    const int a = s->bond_atoms[2 * b], c = s->bond_atoms[2 * b + 1];
    if (a < 0 || a >= n || c < 0 || c >= n || a == c) return fail(e, SCEMA_MD_ERR_ARG, "bad bond %d", b);
    adj[a].push_back(c);
    adj[c].push_back(a);
    max_r0 = std::max(max_r0, s->bond_coeff[2 * s->bond_type[b] + 1]);
  }
  std::vector<int> sp_at;
  std::vector<double> sp_cf;
  std::vector<std::vector<int>> excl(n);
  std::vector<int> level(n, 0), frontier, next, touched;
  int max_excl_level = 0;
  for (int i = 0; i < n; i++) {
    level[i] = -1;
    touched.assign(1, i);
    frontier.assign(1, i);
    for (int lvl = 1; lvl <= 3; lvl++) {
      next.clear();
      for (int a : frontier)
        for (int c : adj[a])
          if (level[c] == 0) {
            level[c] = lvl;
            next.push_back(c);
            touched.push_back(c);
          }
      frontier.swap(next);
    }
    for (int c : touched) {
      if (c > i) {
        const int lvl = level[c];
        const double wl = s->special_lj[lvl - 1], wc = s->special_coul[lvl - 1];
        if (!(wl == 1.0 && wc == 1.0)) {
          sp_at.push_back(i);
          sp_at.push_back(c);
          sp_cf.push_back(wl);
          sp_cf.push_back(wc);
          excl[i].push_back(c);
          excl[c].push_back(i);
          max_excl_level = std::max(max_excl_level, lvl);
        }
      }
    }
    for (int c : touched) level[c] = 0;
  }
  t.nspecial = (int)sp_cf.size() / 2;
  std::vector<int> ex_start(n + 1, 0), ex_list;
  for (int i = 0; i < n; i++) {
    std::sort(excl[i].begin(), excl[i].end());
    ex_start[i + 1] = ex_start[i] + (int)excl[i].size();
    ex_list.insert(ex_list.end(), excl[i].begin(), excl[i].end());
  }
  // build-time exclusion gate: an excluded pair is at most max_excl_level bonds apart
  t.excl_cut = 1.5 * max_excl_level * max_r0;
  // ---- fix shake ... m <mass>: star clusters ----
  std::vector<char> shaken(s->nbonds, 0);
  std::vector<int> nsh(n, 0);
  if (e->p.shake_mass > 0.0)
    for (int b = 0; b < s->nbonds; b++) {
      const int a = s->bond_atoms[2 * b], c = s->bond_atoms[2 * b + 1];
      if (std::fabs(t.mass_atom[a] - e->p.shake_mass) <= 0.1 || std::fabs(t.mass_atom[c] - e->p.shake_mass) <= 0.1) {
        shaken[b] = 1;
        nsh[a]++;
        nsh[c]++;
      }
    }
  std::vector<int> cl_of(n, -1), clus_at, clus_n;
  std::vector<double> clus_d;
  for (int b = 0; b < s->nbonds; b++) {
    if (!shaken[b]) continue;
    const int a = s->bond_atoms[2 * b], c = s->bond_atoms[2 * b + 1];
    int cen, sat;
    if (nsh[a] > nsh[c] || (nsh[a] == nsh[c] && a < c)) { cen = a; sat = c; } else { cen = c; sat = a; }
    if (nsh[sat] != 1) return fail(e, SCEMA_MD_ERR_ARG, "SHAKE cluster is not star shaped (atom %d)", sat);
    int cl = cl_of[cen];
    if (cl < 0) {
      cl = (int)clus_n.size();
      cl_of[cen] = cl;
      clus_n.push_back(1);
      clus_at.insert(clus_at.end(), {cen, 0, 0, 0});
      clus_d.insert(clus_d.end(), {0.0, 0.0, 0.0});
    }
    const int k = clus_n[cl];
    if (k >= 4) return fail(e, SCEMA_MD_ERR_ARG, "SHAKE cluster of more than 4 atoms");
    clus_at[4 * cl + k] = sat;
    clus_d[3 * cl + (k - 1)] = s->bond_coeff[2 * s->bond_type[b] + 1];
    clus_n[cl] = k + 1;
    t.ncons++;
  }
  t.nclus = (int)clus_n.size();
  // the atoms outside the clusters (k_finish walks clusters, then these)
  std::vector<int> free_at;
  {
    std::vector<char> in_cl(n, 0);
    for (int cl = 0; cl < t.nclus; cl++)
      for (int k = 0; k < clus_n[cl]; k++) in_cl[clus_at[4 * cl + k]] = 1;
    for (int i = 0; i < n; i++)
      if (!in_cl[i]) free_at.push_back(i);
  }
  t.nfree = (int)free_at.size();
  // ---- per-term coefficient expansion; unconstrained bonds first ----
  std::vector<int> bond_at, angle_at(s->angle_atoms, s->angle_atoms + 3 * (size_t)s->nangles),
      dih_at(s->dihedral_atoms, s->dihedral_atoms + 4 * (size_t)s->ndihedrals),
      imp_at(s->improper_atoms, s->improper_atoms + 4 * (size_t)s->nimpropers);
  std::vector<double> bond_cf, angle_cf, dih_cf, imp_cf;
  for (int pass = 0; pass < 2; pass++)
    for (int b = 0; b < s->nbonds; b++) {
      if ((int)shaken[b] != pass) continue;
      bond_at.push_back(s->bond_atoms[2 * b]);
      bond_at.push_back(s->bond_atoms[2 * b + 1]);
      bond_cf.push_back(s->bond_coeff[2 * s->bond_type[b]]);
      bond_cf.push_back(s->bond_coeff[2 * s->bond_type[b] + 1]);
      if (pass == 0) t.nbonds_noshake++;
    }
  t.nbonds = s->nbonds;
  t.nangles = s->nangles;
  t.ndihedrals = s->ndihedrals;
  t.nimpropers = s->nimpropers;
  for (int m = 0; m < s->nangles; m++) {
    angle_cf.push_back(s->angle_coeff[2 * s->angle_type[m]]);
    angle_cf.push_back(s->angle_coeff[2 * s->angle_type[m] + 1]);
  }
  for (int m = 0; m < s->ndihedrals; m++)
    for (int k = 0; k < 4; k++) dih_cf.push_back(s->dihedral_coeff[4 * s->dihedral_type[m] + k]);
  for (int m = 0; m < s->nimpropers; m++) {
    imp_cf.push_back(s->improper_coeff[2 * s->improper_type[m]]);
    imp_cf.push_back(s->improper_coeff[2 * s->improper_type[m] + 1]);
  }
  for (size_t k = 0; k < angle_at.size(); k++)
    if (angle_at[k] < 0 || angle_at[k] >= n) return fail(e, SCEMA_MD_ERR_ARG, "bad angle atom");
  for (size_t k = 0; k < dih_at.size(); k++)
    if (dih_at[k] < 0 || dih_at[k] >= n) return fail(e, SCEMA_MD_ERR_ARG, "bad dihedral atom");
  for (size_t k = 0; k < imp_at.size(); k++)
    if (imp_at[k] < 0 || imp_at[k] >= n) return fail(e, SCEMA_MD_ERR_ARG, "bad improper atom");
  // ---- bonded tiles (md_bonded.hip): no atomics, no zeroing ----
  // Atoms are ranked by a breadth-first walk of the bond graph, so that consecutive ranks are topological
  // neighbours whatever the numbering of the input.  A tile = BT_OWNERS consecutive ranks (its owners).  It
  // evaluates EVERY term that touches one of its owners and keeps only the forces on its owners, which it writes
  // with plain coalesced stores (fb is indexed by rank): a term whose atoms span two tiles is evaluated by both
  // (chain molecules cut every few hundred atoms: a few per cent of the terms; none for PE-10k, whose 96-atom
  // rings fill a tile two by two) and counted once for the virial and the energies -- by the tile that owns its
  // lowest-ranked atom (the others carry BT_NOCOUNT on their first atom index).  Positions of the tile's local
  // atoms (owners first, then the halo of up to three bonds) are staged and forces accumulated per tile in LDS.
  std::vector<int> rank(n, -1), by_rank;
  by_rank.reserve(n);
  {
    std::vector<int> queue;
    for (int root = 0; root < n; root++) {
      if (rank[root] >= 0) continue;
      rank[root] = (int)by_rank.size();
      by_rank.push_back(root);
      queue.assign(1, root);
      for (size_t h = 0; h < queue.size(); h++)
        for (int c : adj[queue[h]])
          if (rank[c] < 0) {
            rank[c] = (int)by_rank.size();
            by_rank.push_back(c);
            queue.push_back(c);
          }
    }
  }
  const int ntile = (n + BT_OWNERS - 1) / BT_OWNERS;
  struct TermRef { int kind, idx, count; };
  std::vector<std::vector<TermRef>> tile_terms(ntile);
  auto add_term = [&](const int *atoms, int cnt, int kind, int idx) {
    int rmin = rank[atoms[0]];
    for (int k = 1; k < cnt; k++) rmin = std::min(rmin, rank[atoms[k]]);
    int seen[4], ns_ = 0;
    for (int k = 0; k < cnt; k++) {
      const int tl = rank[atoms[k]] / BT_OWNERS;
      bool dup = false;
      for (int q = 0; q < ns_; q++) dup = dup || seen[q] == tl;
      if (dup) continue;
      seen[ns_++] = tl;
      tile_terms[tl].push_back({kind, idx, tl == rmin / BT_OWNERS ? 1 : 0});
    }
  };
  for (int m = 0; m < s->nbonds; m++) add_term(&bond_at[2 * m], 2, m < t.nbonds_noshake ? BT_BOND : BT_BOND_SHAKEN, m);
  for (int m = 0; m < s->nangles; m++) add_term(&angle_at[3 * m], 3, BT_ANGLE, m);
  for (int m = 0; m < s->ndihedrals; m++) add_term(&dih_at[4 * m], 4, BT_DIHEDRAL, m);
  for (int m = 0; m < s->nimpropers; m++) add_term(&imp_at[4 * m], 4, BT_IMPROPER, m);
  for (int m = 0; m < t.nspecial; m++) add_term(&sp_at[2 * m], 2, BT_SPECIAL, m);
  // Tile-ordered term stream: ONE 64-bit descriptor per term (BT_D_* in md_types.h: four 10-bit local atom indices, the
  // term's type, kind, special-bond level and the count flag); coefficients are looked up by type in small tables the
  // kernel stages in LDS.  Kinds follow each other, each padded to whole chunks of 64 descriptors, so a wave always
  // runs one formula; chunk c of a tile goes to wave c % 4 of its workgroup, which requests all its descriptors with
  // its first instructions: one memory latency per tile instead of one per kind and pass.
  std::vector<int> bond_ty;   // type of the reordered bonds
  for (int pass = 0; pass < 2; pass++)
    for (int b2 = 0; b2 < s->nbonds; b2++)
      if ((int)shaken[b2] == pass) bond_ty.push_back(s->bond_type[b2]);
  std::vector<int> sp_lvl(t.nspecial, 1);
  for (int m = 0; m < t.nspecial; m++)
    for (int lvl = 1; lvl <= 3; lvl++)
      if (sp_cf[2 * m] == s->special_lj[lvl - 1] && sp_cf[2 * m + 1] == s->special_coul[lvl - 1]) { sp_lvl[m] = lvl; break; }
  // coefficient tables: bonds (K, r0), angles (K, theta0), dihedrals (K1..K4), impropers (K, chi0)
  std::vector<double> coef;
  int cf_off[4];
  cf_off[0] = 0;
  coef.insert(coef.end(), s->bond_coeff, s->bond_coeff + 2 * (size_t)s->nbondtypes);
  cf_off[1] = (int)coef.size();
  coef.insert(coef.end(), s->angle_coeff, s->angle_coeff + 2 * (size_t)s->nangletypes);
  cf_off[2] = (int)coef.size();
  coef.insert(coef.end(), s->dihedral_coeff, s->dihedral_coeff + 4 * (size_t)s->ndihedraltypes);
  cf_off[3] = (int)coef.size();
  coef.insert(coef.end(), s->improper_coeff, s->improper_coeff + 2 * (size_t)s->nimpropertypes);
  if (coef.size() > BT_MAXCOEF)
    return fail(e, SCEMA_MD_ERR_ARG, "%zu bonded coefficients (bond/angle/dihedral/improper types): at most %d fit the LDS table of the bonded kernel", coef.size(), BT_MAXCOEF);
  if (s->nbondtypes > BT_D_TMASK + 1 || s->nangletypes > BT_D_TMASK + 1 || s->ndihedraltypes > BT_D_TMASK + 1 || s->nimpropertypes > BT_D_TMASK + 1)
    return fail(e, SCEMA_MD_ERR_ARG, "more than %d types of one bonded kind", BT_D_TMASK + 1);
  for (int k = 0; k < 4; k++) t.bt_cf_off[k] = cf_off[k];
  t.bt_ncoef = (int)coef.size();
  for (int k = 0; k < 3; k++) { t.sp_w[k] = s->special_lj[k]; t.sp_w[3 + k] = s->special_coul[k]; }
  std::vector<int> bt_desc((size_t)ntile * BT_DESC, 0), bt_atoms;
  std::vector<unsigned long long> bt_terms;
  const int natm[BT_NKIND] = {2, 2, 3, 4, 4, 2};
  std::vector<int> local_of(n, -1);
  t.bt_maxloc = 1;
  t.bt_maxchunk = 1;
  for (int tl = 0; tl < ntile; tl++) {
    int *desc = &bt_desc[(size_t)tl * BT_DESC];
    desc[0] = (int)bt_atoms.size();
    // local atom list: the owners in rank order (local index = rank - first rank of the tile), then the halo by rank
    std::vector<int> members, halo;
    const int r0 = tl * BT_OWNERS, r1 = std::min(n, r0 + BT_OWNERS);
    for (int r = r0; r < r1; r++) { members.push_back(by_rank[r]); local_of[by_rank[r]] = r - r0; }
    auto term_atoms = [&](const TermRef &tr) -> const int * {
      const int m = tr.idx;
      return (tr.kind <= BT_BOND_SHAKEN) ? &bond_at[2 * m] : (tr.kind == BT_ANGLE) ? &angle_at[3 * m] : (tr.kind == BT_DIHEDRAL) ? &dih_at[4 * m]
             : (tr.kind == BT_IMPROPER) ? &imp_at[4 * m] : &sp_at[2 * m];
    };
    for (const TermRef &tr : tile_terms[tl]) {
      const int *at = term_atoms(tr);
      for (int k = 0; k < natm[tr.kind]; k++)
        if (local_of[at[k]] < 0) { local_of[at[k]] = 0; halo.push_back(at[k]); }
    }
    std::sort(halo.begin(), halo.end(), [&](int a, int b) { return rank[a] < rank[b]; });
    for (size_t l = 0; l < halo.size(); l++) local_of[halo[l]] = (int)(members.size() + l);
    members.insert(members.end(), halo.begin(), halo.end());
    if ((int)members.size() > BT_D_LMASK + 1)
      return fail(e, SCEMA_MD_ERR_ARG, "a bonded tile touches %zu atoms (more than %d): topology too branched for the tile descriptors", members.size(), BT_D_LMASK + 1);
    desc[2] = (int)(bt_terms.size() / 64);   // first chunk of the tile
    for (int kind = 0; kind < BT_NKIND; kind++) {
      // Terms that follow each other in the input share atoms (the nine torsions around one bond): dealt to
      // consecutive lanes they would hit the same LDS accumulators in the same instruction.  A stride
      // permutation spreads them over the tile instead.
      std::vector<const TermRef *> of_kind;
      for (const TermRef &tr : tile_terms[tl])
        if (tr.kind == kind) of_kind.push_back(&tr);
      const int nk_ = (int)of_kind.size();
      int stride = 1;
      if (nk_ > 16)
        for (stride = 13; stride < nk_; stride += 2) {   // smallest odd stride >= 13 coprime with nk_
          int a_ = stride, b_ = nk_;
          while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
          if (a_ == 1) break;
        }
      if (stride >= nk_) stride = 1;
      for (int q = 0; q < nk_; q++) {
        const TermRef &tr = *of_kind[(int)(((long long)q * stride) % std::max(nk_, 1))];
        const int m = tr.idx;
        const int *at = term_atoms(tr);
        unsigned long long d = BT_D_VALID | ((unsigned long long)kind << BT_D_KIND_SHIFT);
        for (int k = 0; k < natm[kind]; k++) d |= (unsigned long long)local_of[at[k]] << (10 * k);
        const int ty = (kind <= BT_BOND_SHAKEN) ? bond_ty[m] : (kind == BT_ANGLE) ? s->angle_type[m] : (kind == BT_DIHEDRAL) ? s->dihedral_type[m]
                       : (kind == BT_IMPROPER) ? s->improper_type[m] : 0;
        d |= (unsigned long long)ty << BT_D_TYPE_SHIFT;
        if (kind == BT_SPECIAL) d |= (unsigned long long)sp_lvl[m] << BT_D_LVL_SHIFT;
        if (!tr.count) d |= BT_D_NOCOUNT;
        bt_terms.push_back(d);
      }
      while (bt_terms.size() % 64) bt_terms.push_back(0ull);   // whole chunks per kind (an invalid descriptor is all zero)
    }
    desc[3] = (int)(bt_terms.size() / 64) - desc[2];   // chunks of the tile
    desc[1] = (int)members.size();
    desc[14] = r1 - r0;   // owners
    t.bt_maxloc = std::max(t.bt_maxloc, (int)members.size());
    t.bt_maxchunk = std::max(t.bt_maxchunk, desc[3]);
    for (int atom : members) { bt_atoms.push_back(atom); local_of[atom] = -1; }
  }
  t.bt_ntile = ntile;
  t.original.take(*s);
  std::memcpy(t.init_box, s->box, sizeof t.init_box);
  t.init_x.assign(s->x, s->x + 3 * (size_t)n);
  t.init_v.assign(s->v, s->v + 3 * (size_t)n);
  int rc;
  if ((rc = upload(e, t.d_type, t.type))) return rc;
  if ((rc = upload(e, t.d_q, t.q))) return rc;
  if ((rc = upload(e, t.d_mass, t.mass_atom))) return rc;
  if ((rc = upload(e, t.d_lj, t.lj))) return rc;
  if ((rc = upload(e, t.d_bt_terms, bt_terms))) return rc;
  if ((rc = upload(e, t.d_bt_coef, coef))) return rc;
  if ((rc = upload(e, t.d_bt_desc, bt_desc))) return rc;
  if ((rc = upload(e, t.d_bt_atoms, bt_atoms))) return rc;
  if ((rc = upload(e, t.d_bt_rank, rank))) return rc;
  if ((rc = upload(e, t.d_ex_start, ex_start))) return rc;
  if ((rc = upload(e, t.d_ex_list, ex_list))) return rc;
  if ((rc = upload(e, t.d_clus_at, clus_at))) return rc;
  if ((rc = upload(e, t.d_clus_n, clus_n))) return rc;
  if ((rc = upload(e, t.d_clus_d, clus_d))) return rc;
  if ((rc = upload(e, t.d_free_at, free_at))) return rc;
  return SCEMA_MD_OK;
}


}  // namespace scema_eng
