// engine_comm.cpp -- one process per GPU: communicator, agreement handshake, state migration, the stress all-gather (replaces stmd_sync.h:620-726), the planner's C face
#include "engine.h"
#include "../md_env.h"

namespace scema_eng {


// Replica states change GPU (scema::PlanMove): x, v and the box of the source state of simulation m.sim go from rank
// m.from to rank m.to, where they become the state that simulation continues from.  The source rank keeps its copy when
// another quadrature point branches from it (most_recent_qp_id != qp_id); a state that merely moved is dropped there after
// the update.  RCCL: one group of point-to-point sends/receives over xGMI on the engine's stream; host transport: the
// moves in plan order, blocking send/recv pairs (every rank walks the same list, so the pairs cannot cross).
//
// An update is TWO collectives: the 16-byte agreement handshake and the stress all-gather (with the exchange of states between
// them where the plan has moves).  Everything that can fail on ONE rank alone happens before the handshake, in prepare_incoming:
// the receiving rank's allocations, the look-up of every source state this rank must hold, the device buffer of the boxes that
// travel beside x and v and its upload -- a failure there is the rank's status word in the handshake and ends the call on every
// rank before anything is posted.  migrate_states runs after the ranks agreed to exchange and works on what prepare_incoming
// resolved (pointers, no look-ups, no allocations): no path returns between the agreement and the end of the exchange.  A call
// that fails inside the RCCL group does not stop the queueing -- every remaining send and receive of this rank is still posted,
// so every operation a peer has posted finds its partner --, the group is ended whatever happened inside it, the host transport
// posts every send and receive of its moves, and the error is RETURNED for the status word of the stress all-gather: the other
// ranks are never left inside a collective this rank skipped.
constexpr int MIG_SIDE = 10;   // doubles that travel next to x and v of a migrating state: box[9], State::skin_extra

// test hook SCEMA_MD_TEST_FAIL_MIGRATE = "<what>[:<rank>]" (rank omitted or negative: every rank): what = dbox (the device buffer of
// the boxes cannot be allocated: before the handshake), upload (its host-to-device copy fails: before the handshake), enqueue (the
// first point-to-point call of the exchange reports a failure -- it is posted all the same, as a call that failed AFTER being
// queued would be, so that the peers can finish --), group (ncclGroupEnd reports a failure), hostcopy (the first device copy of
// the host transport fails)
static bool inject_migrate_failure(const char *what, int rank) {
  const char *v = scema_env("SCEMA_MD_TEST_FAIL_MIGRATE");
  if (!v) return false;
  const size_t n = std::strlen(what);
  if (std::strncmp(v, what, n) != 0 || (v[n] != 0 && v[n] != ':')) return false;
  if (v[n] == 0) return true;
  const int r = atoi(v + n + 1);
  return r < 0 || r == rank;
}

// what this rank needs for the exchange, made (and checked) before the handshake: a failure here must travel in its status word
int prepare_incoming(scema_md_engine *e, const scema_mdsim *sims, const scema::SimPlan &plan, const std::vector<std::string> &src_keys,
                     std::map<int, std::unique_ptr<State>> &incoming) {
  Comm &c = e->comm;
  const int rank = c.rank;
  const int nm = (int)plan.moves.size();
  c.mig_src.assign(nm, nullptr);
  c.mig_dst.assign(nm, nullptr);
  c.h_box.assign(MIG_SIDE * (size_t)nm, 0.0);   // box[9] + the state's list skin (State::skin_extra)
  if (c.kind == 2 && nm > 0 && (!c.send || !c.recv))
    return fail(e, SCEMA_MD_ERR_ARG, "the host communicator has no send/recv callbacks: replica states cannot move between ranks");
  for (int k = 0; k < nm; k++) {
    const scema::PlanMove &m = plan.moves[k];
    if (m.from == rank) {
      auto it = e->states.find(src_keys[m.sim]);
      if (it == e->states.end())
        return fail(e, SCEMA_MD_ERR_NOSTATE, "rank %d is recorded as the owner of state %s but does not hold it", rank, src_keys[m.sim].c_str());
      c.mig_src[k] = it->second.get();
      std::memcpy(&c.h_box[MIG_SIDE * (size_t)k], c.mig_src[k]->box, 9 * sizeof(double));
      c.h_box[MIG_SIDE * (size_t)k + 9] = c.mig_src[k]->skin_extra;
    }
    if (m.to != rank) continue;
    Topo *t = find_topo(e, sims[m.sim].matid, sims[m.sim].replica);
    if (!t) return fail(e, SCEMA_MD_ERR_NOSTATE, "replica %s_%d is not registered on rank %d", sims[m.sim].matid, sims[m.sim].replica, rank);
    if (const char *fr = scema_env("SCEMA_MD_TEST_FAIL_INCOMING"))   // test hook: the allocation fails on that rank
      if (atoi(fr) == rank || atoi(fr) < 0) return fail(e, SCEMA_MD_ERR_DEVICE, "out of device memory for a replica state that migrates to rank %d (injected: SCEMA_MD_TEST_FAIL_INCOMING)", rank);
    if (int rc = make_empty_state(e, t, incoming[m.sim])) {
      (void)hipGetLastError();   // (the allocation's error is reported, not left for the next launch to find)
      return fail(e, rc, "out of device memory for a replica state that migrates to rank %d", rank);
    }
    c.mig_dst[k] = incoming[m.sim].get();
  }
  if (c.kind == 1 && nm > 0) {
    // the boxes travel through a device buffer: allocated and filled here, where a failure still reaches every rank
    hipError_t rc = inject_migrate_failure("dbox", rank) ? hipErrorOutOfMemory : c.d_box.ensure(2 * c.h_box.size() * sizeof(double));   // (second half: where a state sent to this very rank receives its box)
    if (rc != hipSuccess) {
      (void)hipGetLastError();
      return fail(e, SCEMA_MD_ERR_DEVICE, "out of device memory for the boxes of the replica states that migrate (rank %d): %s", rank, hipGetErrorString(rc));
    }
    rc = inject_migrate_failure("upload", rank) ? hipErrorInvalidValue : hipMemcpyAsync(c.d_box.p, c.h_box.data(), c.h_box.size() * sizeof(double), hipMemcpyHostToDevice, e->stream);
    if (rc != hipSuccess) {
      (void)hipGetLastError();
      return fail(e, SCEMA_MD_ERR_DEVICE, "upload of the boxes of the replica states that migrate failed on rank %d: %s", rank, hipGetErrorString(rc));
    }
  }
  return SCEMA_MD_OK;
}

int migrate_states(scema_md_engine *e, const scema::SimPlan &plan) {
  Comm &c = e->comm;
  const int nm = (int)plan.moves.size();
  int err = SCEMA_MD_OK;
  if ((int)c.mig_src.size() != nm || (int)c.mig_dst.size() != nm)   // (prepare_incoming ran for this plan: a broken call order is a bug of the library, found on every rank alike)
    return fail(e, SCEMA_MD_ERR_ARG, "internal: the exchange of replica states was not prepared for this plan");
  if (c.kind == 1) {
    bool first = true;
    // inside the group nothing returns and nothing stops the queueing: a failing call is noted, the rest is posted all the same
    auto q = [&](ncclResult_t r, const char *what, int peer) {
      if (first && inject_migrate_failure("enqueue", c.rank)) r = ncclInternalError;
      first = false;
      if (r != ncclSuccess && !err) err = fail(e, SCEMA_MD_ERR_DEVICE, "%s with rank %d failed while replica states migrate: %s", what, peer, ncclGetErrorString(r));
    };
    ncclResult_t gs = ncclGroupStart();
    if (gs != ncclSuccess) err = fail(e, SCEMA_MD_ERR_DEVICE, "ncclGroupStart failed while replica states migrate: %s", ncclGetErrorString(gs));
    for (int k = 0; k < nm; k++) {
      const scema::PlanMove &m = plan.moves[k];
      double *dbox = c.d_box.as<double>() + MIG_SIDE * (size_t)k;
      if (c.rank == m.from && c.mig_src[k]) {
        State *s = c.mig_src[k];
        const size_t cnt = 3 * (size_t)s->topo->natoms;
        q(ncclSend(s->x.p, cnt, ncclDouble, m.to, c.nccl, e->stream), "ncclSend", m.to);
        q(ncclSend(s->v.p, cnt, ncclDouble, m.to, c.nccl, e->stream), "ncclSend", m.to);
        q(ncclSend(dbox, MIG_SIDE, ncclDouble, m.to, c.nccl, e->stream), "ncclSend", m.to);
      }
      if (c.rank == m.to && c.mig_dst[k]) {
        State *d = c.mig_dst[k];
        const size_t cnt = 3 * (size_t)d->topo->natoms;
        // (a state sent to this very rank -- the one-rank test of these calls -- receives its box behind the sent ones)
        double *rbox = (m.from == m.to) ? c.d_box.as<double>() + MIG_SIDE * (size_t)(nm + k) : dbox;
        q(ncclRecv(d->x.p, cnt, ncclDouble, m.from, c.nccl, e->stream), "ncclRecv", m.from);
        q(ncclRecv(d->v.p, cnt, ncclDouble, m.from, c.nccl, e->stream), "ncclRecv", m.from);
        q(ncclRecv(rbox, MIG_SIDE, ncclDouble, m.from, c.nccl, e->stream), "ncclRecv", m.from);
      }
    }
    if (gs == ncclSuccess) {
      ncclResult_t ge = ncclGroupEnd();
      if (inject_migrate_failure("group", c.rank)) ge = ncclInternalError;
      if (ge != ncclSuccess && !err) err = fail(e, SCEMA_MD_ERR_DEVICE, "ncclGroupEnd failed while replica states migrate: %s", ncclGetErrorString(ge));
    }
    // (from here on the exchange is over for the peers: what fails now is this rank's own business and travels in its status word)
    if (err) return err;
    std::vector<double> hb(2 * c.h_box.size(), 0.0);
    HIPCHK(hipMemcpyAsync(hb.data(), c.d_box.p, hb.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int k = 0; k < nm; k++)
      if (c.rank == plan.moves[k].to && c.mig_dst[k]) {
        const double *b = &hb[MIG_SIDE * (size_t)(plan.moves[k].from == plan.moves[k].to ? nm + k : k)];
        std::memcpy(c.mig_dst[k]->box, b, 9 * sizeof(double));
        c.mig_dst[k]->skin_extra = b[9];
      }
  } else {
    // (the callbacks are checked when the communicator is attached and again before the handshake, scema_md_strain_batch)
    // a device copy that fails does not stop the walk: the peer of every move is waiting in its own send or receive
    bool first = true;
    auto dev = [&](hipError_t r, const char *what) {
      if (first && inject_migrate_failure("hostcopy", c.rank)) r = hipErrorInvalidValue;
      first = false;
      if (r != hipSuccess && !err) err = fail(e, SCEMA_MD_ERR_DEVICE, "%s failed while replica states migrate: %s", what, hipGetErrorString(r));
    };
    std::vector<double> buf;
    for (int k = 0; k < nm; k++) {
      const scema::PlanMove &m = plan.moves[k];
      if (c.rank != m.from && c.rank != m.to) continue;
      State *st = (c.rank == m.from) ? c.mig_src[k] : c.mig_dst[k];
      if (!st) continue;
      const size_t n3 = 3 * (size_t)st->topo->natoms;
      buf.assign(2 * n3 + MIG_SIDE, 0.0);
      if (c.rank == m.from) {
        dev(hipMemcpyAsync(buf.data(), st->x.p, n3 * 8, hipMemcpyDeviceToHost, e->stream), "copy of a state to the host");
        dev(hipMemcpyAsync(buf.data() + n3, st->v.p, n3 * 8, hipMemcpyDeviceToHost, e->stream), "copy of a state to the host");
        dev(hipStreamSynchronize(e->stream), "copy of a state to the host");
        std::memcpy(buf.data() + 2 * n3, st->box, 9 * sizeof(double));
        buf[2 * n3 + 9] = st->skin_extra;
        if ((!c.send || c.send(c.ctx, buf.data(), (int64_t)(buf.size() * 8), m.to)) && !err) err = fail(e, SCEMA_MD_ERR_DEVICE, "host send of a replica state to rank %d failed", m.to);
      } else {
        if ((!c.recv || c.recv(c.ctx, buf.data(), (int64_t)(buf.size() * 8), m.from)) && !err) err = fail(e, SCEMA_MD_ERR_DEVICE, "host receive of a replica state from rank %d failed", m.from);
        dev(hipMemcpyAsync(st->x.p, buf.data(), n3 * 8, hipMemcpyHostToDevice, e->stream), "copy of a state to the device");
        dev(hipMemcpyAsync(st->v.p, buf.data() + n3, n3 * 8, hipMemcpyHostToDevice, e->stream), "copy of a state to the device");
        dev(hipStreamSynchronize(e->stream), "copy of a state to the device");
        std::memcpy(st->box, buf.data() + 2 * n3, 9 * sizeof(double));
        st->skin_extra = buf[2 * n3 + 9];
      }
    }
    if (err) return err;
  }
  c.migrations += nm;
  return SCEMA_MD_OK;
}

// Result buffer of a rank: 6*cap stresses followed by SCEMA_MD_RESULT_TRAILER words -- [status of this rank's share
// (0 = fine, else the error code), hash of the plan this rank computed].  Every rank enters the collective whatever
// happened to its share, so that a rank-local failure (list overflow, a replica that blew up, a missing state) ends the
// update on ALL ranks together instead of leaving the others blocked in the collective.
static_assert(SCEMA_MD_RESULT_TRAILER == 2, "result trailer: status, plan hash");

// 52 bits of an FNV-1a hash: exactly representable in the double it travels in
double plan_hash(const scema::SimPlan &P, const std::vector<double> &cost) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) {
    for (int k = 0; k < 8; k++) { h ^= (v >> (8 * k)) & 0xffull; h *= 1099511628211ull; }
  };
  mix((unsigned long long)P.world); mix((unsigned long long)P.cap); mix(P.owner.size());
  for (size_t i = 0; i < P.owner.size(); i++) {
    mix((unsigned long long)P.owner[i]); mix((unsigned long long)P.pos[i]); mix((unsigned long long)(long long)P.home[i]);
    unsigned long long bits; std::memcpy(&bits, &cost[i], 8); mix(bits);
  }
  mix(P.moves.size());
  for (const scema::PlanMove &m : P.moves) { mix((unsigned long long)m.sim); mix((unsigned long long)m.from); mix((unsigned long long)m.to); }
  return (double)(h & ((1ull << 52) - 1));
}

// every rank contributes cnt doubles (host memory), out = world * cnt
int comm_allgather(scema_md_engine *e, const double *local, size_t cnt, std::vector<double> &out, DevBuf &d_send) {
  Comm &c = e->comm;
  out.assign(cnt * c.world, 0.0);
  if (c.kind == 1) {
    HIPCHK(d_send.ensure(cnt * sizeof(double)));
    HIPCHK(c.d_gather.ensure(cnt * c.world * sizeof(double)));
    HIPCHK(hipMemcpyAsync(d_send.p, local, cnt * sizeof(double), hipMemcpyHostToDevice, e->stream));
    NCCLCHK(ncclAllGather(d_send.p, c.d_gather.p, cnt, ncclDouble, c.nccl, e->stream));
    HIPCHK(hipMemcpyAsync(out.data(), c.d_gather.p, cnt * c.world * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else {
    if (!c.ag || c.ag(c.ctx, local, out.data(), (int64_t)(cnt * sizeof(double))))
      return fail(e, SCEMA_MD_ERR_DEVICE, "host all-gather failed");
  }
  return SCEMA_MD_OK;
}

// what the ranks told each other: the first failing rank's code, or a plan mismatch
int check_gathered_trailers(scema_md_engine *e, const double *gathered, size_t stride, size_t off, int world, int rank, const char *when) {
  for (int r = 0; r < world; r++) {
    const int st = (int)gathered[r * stride + off];
    if (st != 0) {
      if (r == rank) return st;   // this rank's own message is already in e->err
      return fail(e, st, "rank %d failed %s (code %d): the update is abandoned on every rank", r, when, st);
    }
  }
  for (int r = 1; r < world; r++)
    if (gathered[r * stride + off + 1] != gathered[off + 1])
      return fail(e, SCEMA_MD_ERR_ARG, "ranks 0 and %d computed different plans for this update: replicas, states and request vectors must be the same on every rank "
                  "(scema_md_register_replica / set_state / drop_state / load_state_file / equilibrate are collective when a world > 1 is used)", r);
  return SCEMA_MD_OK;
}

// Agreement before anything moves: 2 doubles per rank (status of the local pre-checks, plan hash).  A plan that differs
// between ranks would pair sends with no receive, or give the all-gather different counts.
int handshake(scema_md_engine *e, int local_status, double hash) {
  Comm &c = e->comm;
  const double word[2] = {(double)local_status, hash};
  std::vector<double> all;
  int rc = comm_allgather(e, word, 2, all, c.d_word);
  if (rc) return rc;
  c.handshakes += 1;
  return check_gathered_trailers(e, all.data(), 2, 0, c.world, c.rank, "before the update started");
}

// ONE all-gather of 6*cap (+ trailer) doubles per rank, then every rank fills every sims[i].stress (all ranks hold all
// stresses, so the second share_scale_bridging_data broadcast of the caller, dealammps.cc:458, is not needed).
int allgather_stresses(scema_md_engine *e, const std::vector<double> &local, scema_mdsim *sims, int n_sims) {
  Comm &c = e->comm;
  const scema::SimPlan &plan = e->last_plan;
  const size_t cnt = local.size();
  int rc = comm_allgather(e, local.data(), cnt, c.h_gather, e->d_local_stress);
  if (rc) return rc;
  c.allgathers += 1;
  rc = check_gathered_trailers(e, c.h_gather.data(), cnt, cnt - SCEMA_MD_RESULT_TRAILER, c.world, c.rank, "during the update");
  if (rc) return rc;
  for (int i = 0; i < n_sims; i++) {
    const double *src = c.h_gather.data() + ((size_t)plan.owner[i] * cnt + 6 * (size_t)plan.pos[i]);
    for (int k = 0; k < 6; k++) sims[i].stress[k] = src[k];
    sims[i].stress_updated = 1;
  }
  return SCEMA_MD_OK;
}

}  // namespace scema_eng

extern "C" {

// ---- communicator (one process per GPU) ----
int scema_md_comm_unique_id(void *id) {
  if (!id) return SCEMA_MD_ERR_ARG;
  static_assert(SCEMA_MD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return SCEMA_MD_ERR_DEVICE;
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return SCEMA_MD_OK;
}

int scema_md_comm_init_rccl(scema_md_engine *e, const void *id, int32_t rank, int32_t world) {
  if (!e || !id || world <= 0 || rank < 0 || rank >= world) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  if (e->comm.kind) return fail(e, SCEMA_MD_ERR_ARG, "a communicator is already attached");
  (void)settle_pending(e, false);   // (an update of the communicator-less transport that is still waiting for its verdict stands)
  HIPCHK(hipSetDevice(e->p.device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  NCCLCHK(ncclCommInitRank(&e->comm.nccl, world, u, rank));
  e->comm.kind = 1;
  e->comm.rank = rank;
  e->comm.world = world;
  return SCEMA_MD_OK;
}

int scema_md_comm_init_host(scema_md_engine *e, int32_t rank, int32_t world, scema_md_host_allgather_fn allgather, scema_md_host_send_fn send,
                            scema_md_host_recv_fn recv, void *ctx) {
  if (!e || !allgather || world <= 0 || rank < 0 || rank >= world) return fail(e, SCEMA_MD_ERR_ARG, "bad arguments");
  if (e->comm.kind) return fail(e, SCEMA_MD_ERR_ARG, "a communicator is already attached");
  (void)settle_pending(e, false);
  e->comm.kind = 2;
  e->comm.rank = rank;
  e->comm.world = world;
  e->comm.ag = allgather;
  e->comm.send = send;
  e->comm.recv = recv;
  e->comm.ctx = ctx;
  return SCEMA_MD_OK;
}

void scema_md_comm_destroy(scema_md_engine *e) {
  if (!e || !e->comm.kind) return;
  (void)hipSetDevice(e->p.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->comm.kind == 1 && e->comm.nccl) (void)ncclCommDestroy(e->comm.nccl);
  e->comm.nccl = nullptr;
  e->comm.kind = 0;
  e->comm.rank = 0;
  e->comm.world = 1;
  e->comm.ag = nullptr; e->comm.send = nullptr; e->comm.recv = nullptr; e->comm.ctx = nullptr;
}

int32_t scema_md_comm_world(const scema_md_engine *e) { return (e && e->comm.kind) ? e->comm.world : 1; }
int32_t scema_md_comm_rank(const scema_md_engine *e) { return (e && e->comm.kind) ? e->comm.rank : 0; }

int64_t scema_md_comm_handshakes(const scema_md_engine *e) { return e ? e->comm.handshakes : 0; }

int scema_md_comm_stats(const scema_md_engine *e, int64_t *allgathers, int64_t *migrations) {
  if (!e) return SCEMA_MD_ERR_ARG;
  if (allgathers) *allgathers = e->comm.allgathers;
  if (migrations) *migrations = e->comm.migrations;
  return SCEMA_MD_OK;
}

// the recorded owner of a state: rank, or -1 when no rank is recorded (the state, if it exists, is held locally)
int32_t scema_md_state_owner(const scema_md_engine *e, int32_t qp_id, const char *matid, int32_t replica) {
  return e ? e->dir.owner_of(state_key(qp_id, matid, replica)) : -1;
}

// ---- the planner alone: pure host arithmetic (host/sim_plan.h), no GPU needed ----
struct scema_plan_dir {
  scema::OwnerDirectory dir;
};
scema_plan_dir *scema_plan_dir_create(void) { return new scema_plan_dir(); }
void scema_plan_dir_destroy(scema_plan_dir *d) { delete d; }
int scema_plan_update(scema_plan_dir *d, const scema_mdsim *sims, int32_t n_sims, const double *cost, int32_t world, int32_t *owner, int32_t *pos,
                      int32_t *cap, int32_t *moves, int32_t *n_moves, int32_t commit) {
  if (!d || (!sims && n_sims > 0) || n_sims < 0 || world <= 0) return SCEMA_MD_ERR_ARG;
  std::vector<std::string> src(n_sims), dst(n_sims);
  std::vector<double> c(n_sims, 1.0);
  for (int i = 0; i < n_sims; i++) {
    dst[i] = state_key(sims[i].qp_id, sims[i].matid, sims[i].replica);
    if (sims[i].most_recent_qp_id == sims[i].qp_id) src[i] = dst[i];
    else if (sims[i].most_recent_qp_id != SCEMA_MD_QP_NONE) src[i] = state_key(sims[i].most_recent_qp_id, sims[i].matid, sims[i].replica);
    if (cost) c[i] = cost[i];
  }
  const scema::SimPlan P = d->dir.plan(src, dst, c, world);
  for (int i = 0; i < n_sims; i++) {
    if (owner) owner[i] = P.owner[i];
    if (pos) pos[i] = P.pos[i];
  }
  if (cap) *cap = P.cap;
  if (n_moves) *n_moves = (int32_t)P.moves.size();
  if (moves)
    for (size_t k = 0; k < P.moves.size(); k++) { moves[3 * k] = P.moves[k].sim; moves[3 * k + 1] = P.moves[k].from; moves[3 * k + 2] = P.moves[k].to; }
  if (commit) d->dir.commit(P, dst);
  return SCEMA_MD_OK;
}

}  // extern "C"
