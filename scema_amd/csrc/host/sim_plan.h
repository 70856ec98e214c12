// sim_plan.h -- which rank (GPU) runs which simulation of one STMDSync::update(), and which replica states have to
// move for it.
//
// The reference deals simulation i to batch i % n_md_batches (headers/stmd_sync.h:583) and can do so because every
// rank reads last.<qp>.<mat>_<rep>.dump from a shared file system (stmd_problem.h:117-138).  Here a state lives in the
// HBM of ONE GPU, and the update_list changes between updates (FE_problem.h:1330-1350 lists only the quadrature points
// that need MD), so ownership must follow the state, not the position in the request vector:
//
//   * every rank keeps the same OwnerDirectory (state key -> rank); it is a pure function of the request vectors seen
//     so far, so all ranks compute identical plans without talking to each other;
//   * a simulation whose source state (most_recent_qp_id or its own, stmd_problem.h:116-120) already lives on a rank
//     stays there; simulations without a stored state are dealt longest-processing-time-first onto the least loaded
//     rank (ties: round robin from i % world, which reproduces the reference's map for a fresh balanced batch);
//   * then load is levelled: while the most loaded rank holds a simulation cheaper than its lead over the least
//     loaded one, the best such simulation moves (its state travels: PlanMove).  Cost = MD steps (nts + nss), so a
//     ragged batch (nts 10..100) is balanced, a balanced one never migrates.
//
// Pure host C++ (no HIP): the engine (engine/engine_batch.cpp) and the Hooke-mode path of STMDSync use the same planner, and
// the CPU tests drive it through the C ABI (scema_plan_* in include/scema_md.h).
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace scema {

struct PlanMove {
  int sim;       // index into the request vector
  int from, to;  // the source state of that simulation travels from -> to
};

struct SimPlan {
  int world = 1;
  std::vector<int> owner;  // per simulation: the rank that runs it
  std::vector<int> home;   // per simulation: the rank recorded as holding its source state, -1 = none (the directory's view)
  std::vector<int> pos;    // per simulation: slot in its owner's result buffer (6 doubles each)
  std::vector<int> count;  // per rank: simulations it runs
  int cap = 0;             // max count = slots per rank in the one all-gather
  std::vector<PlanMove> moves;
};

class OwnerDirectory {
 public:
  // -1: not recorded = available on every rank (the registered init state, or a state every rank loaded from a file)
  int owner_of(const std::string &key) const {
    auto it = owner_.find(key);
    return it == owner_.end() ? -1 : it->second;
  }
  void erase(const std::string &key) { owner_.erase(key); }
  void clear() { owner_.clear(); }
  void erase_suffix(const std::string &suffix) {   // every state of one (material, replica)
    for (auto it = owner_.begin(); it != owner_.end();) {
      const std::string &k = it->first;
      if (k.size() >= suffix.size() && k.compare(k.size() - suffix.size(), suffix.size(), suffix) == 0) it = owner_.erase(it);
      else ++it;
    }
  }
  size_t size() const { return owner_.size(); }

  SimPlan plan(const std::vector<std::string> &src_keys, const std::vector<std::string> &dst_keys, const std::vector<double> &cost,
               int world) const {
    const int n = (int)dst_keys.size();
    SimPlan P;
    P.world = world;
    P.owner.assign(n, 0);
    P.pos.assign(n, 0);
    P.count.assign(world, 0);
    std::vector<int> &home = P.home;
    home.assign(n, -1);
    if (world > 1) {
      std::vector<double> load(world, 0.0);
      auto least = [&](int start) {   // least loaded rank, ties resolved round robin from `start`
        int best = start % world;
        for (int k = 1; k < world; k++) {
          const int r = (start + k) % world;
          if (load[r] < load[best]) best = r;
        }
        return best;
      };
      for (int i = 0; i < n; i++) {
        home[i] = owner_of(src_keys[i]);
        if (home[i] >= world) home[i] = -1;
        P.owner[i] = home[i];
        if (home[i] >= 0) load[home[i]] += cost[i];
      }
      std::vector<int> order;
      for (int i = 0; i < n; i++)
        if (home[i] < 0) order.push_back(i);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
      for (int i : order) {
        const int r = least(i);
        P.owner[i] = r;
        load[r] += cost[i];
      }
      // levelling: every move lowers sum(load^2), so the loop ends; capped anyway
      for (int iter = 0; iter < 4 * n + 16; iter++) {
        int a = 0, b = 0;
        for (int r = 1; r < world; r++) {
          if (load[r] > load[a]) a = r;
          if (load[r] < load[b]) b = r;
        }
        const double lead = load[a] - load[b];
        int pick = -1;
        double pick_gain = 0.0;
        bool pick_free = false;
        for (int i = 0; i < n; i++) {
          if (P.owner[i] != a || !(cost[i] < lead)) continue;
          // closest to lead/2 levels best; a simulation without a stored state moves for free and wins ties
          const double gain = cost[i] * (lead - cost[i]);
          const bool is_free = home[i] < 0;
          if (pick < 0 || gain > pick_gain * (1.0 + 1e-12) || (gain >= pick_gain * (1.0 - 1e-12) && is_free && !pick_free)) {
            pick = i;
            pick_gain = gain;
            pick_free = is_free;
          }
        }
        if (pick < 0) break;
        P.owner[pick] = b;
        load[a] -= cost[pick];
        load[b] += cost[pick];
      }
      for (int i = 0; i < n; i++)
        if (home[i] >= 0 && home[i] != P.owner[i]) P.moves.push_back({i, home[i], P.owner[i]});
    }
    for (int i = 0; i < n; i++) P.pos[i] = P.count[P.owner[i]]++;
    for (int r = 0; r < world; r++) P.cap = std::max(P.cap, P.count[r]);
    return P;
  }

  // after the update: every simulation's state is stored under its own key on the rank that ran it
  void commit(const SimPlan &P, const std::vector<std::string> &dst_keys) {
    if (P.world <= 1) return;   // a single rank owns everything: nothing to record
    for (size_t i = 0; i < dst_keys.size(); i++) owner_[dst_keys[i]] = P.owner[i];
  }

 private:
  std::map<std::string, int> owner_;
};

}  // namespace scema
