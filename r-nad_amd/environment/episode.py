"""States / Episodes / Buffer -- drop-in for reference environment/episode.py, running on librnad_hip.so.

API kept (names, argument order, attribute names, shapes): `States(tree, B).observations() / .step(actions) /
.indices / .player_to_move / .terminal`; `Episodes(tree, B).generate(net) / .sample(n) / Episodes.collate(list)`
with attributes `t_eff, turns, indices, observations, policy, actions, rewards, values, masks, q_estimates,
v_estimates, generation_time, finished`; `Buffer(max_size).sample / append / clear`.

What differs from the reference (all documented in INTEGRATION.md):
  * state ids and action ids are int32 on the device (reference: int64); `indices` is `[T, B]` int32;
  * `Episodes.generate` writes straight into preallocated `[T_cap, B, ...]` buffers (no per-step clones, list
    appends or torch.stack, episode.py:196-227), runs `T_cap = 2 * tree depth` steps without any host sync and reads
    the per-step alive counters ONCE at the end to trim to the reference's T (episode.py:194 loops until every lane is
    absorbed);
  * `turns`, `actions` (one-hot), `masks`, `q_estimates`, `v_estimates` are materialised lazily from the compact
    primaries (`t & 1`, `action_idx` int32, `mask_bits` u8 / a strided view of `observations`);
  * randomness: torch.multinomial's global sequential generator cannot feed 2^20 lanes, so every decision is a function of
    (seed, global lane, env step): a counter-based uniform (one philox call per game transition) turned into a category by
    the inverse CDF (include/rnad_rng.h) -- one draw from Cat(p) per lane, as torch.multinomial(p, 1) gives.  Explicit
    Exp(1) noise (`noise_action`, `noise_chance`) goes through torch's own algorithm instead, `argmax(p / q)`: with the
    noise the reference consumed, the reference's episodes.  `seed` defaults to a draw from torch's generator, so
    `torch.manual_seed` still makes runs repeatable.
"""
import random
import os
import time
from collections import deque

import numpy
import torch

import rnad_hip
from environment.tree import Tree
from nn.net import MLP as _MLP


def _draw_seed():
    return int(torch.randint(0, 2**62, (1,)).item())


class States:
    """A parallel collection of states of one Tree (reference episode.py:18-125)."""

    def __init__(self, tree: Tree, batch_size, seed=None, lane_offset=0):
        self.tree = tree
        self.batch_size = batch_size
        self._indices = self._player_to_move = None  # created on first use: a native rollout never touches them
        self.row_actions = None
        self.col_actions = None
        self._player = 0  # host copy: the reference itself only ever looks at player_to_move[0] (episode.py:96-98)
        self._step = 0
        self._terminal = False
        self._terminal_stale = False
        self.seed = _draw_seed() if seed is None else int(seed)
        self.lane_offset = int(lane_offset)

    @property
    def indices(self):
        """[B] int32 state ids, all at the root (1) to start with (episode.py:22)."""
        if self._indices is None:
            final = getattr(self, "_final_of", None)
            if final is not None:  # the states a compact rollout ended in: row T of its (lazily rebuilt) indices
                self._indices = final[0].indices[final[1]]
            else:
                self._indices = torch.ones((self.batch_size,), dtype=torch.int32, device=self.tree.device)
        return self._indices

    @indices.setter
    def indices(self, value):
        self._indices = value

    @property
    def player_to_move(self):
        if self._player_to_move is None:
            self._player_to_move = torch.full((self.batch_size,), self._player, dtype=torch.long, device=self.tree.device)
        return self._player_to_move

    @player_to_move.setter
    def player_to_move(self, value):
        self._player_to_move = value

    @property
    def terminal(self):
        """True when every lane sits in the absorbing state 0 (episode.py:124); synchronises when read."""
        if self._terminal_stale:
            self._terminal = bool((self.indices == 0).all().item())
            self._terminal_stale = False
        return self._terminal

    @terminal.setter
    def terminal(self, value):
        self._terminal, self._terminal_stale = bool(value), False

    def observations(self, half=False) -> torch.Tensor:
        """[B, 2, A, A]: expected-value matrix + legal mask from the mover's point of view (episode.py:62-68)."""
        return rnad_hip.observe(self.tree.handle(), self.indices, self._player, half=half)

    def observations_noisy(self):
        return None  # reference placeholder (episode.py:70-82)

    def step(self, actions: torch.Tensor, noise=None) -> torch.Tensor:
        """Commit the mover's actions; on the column player's turn sample chance and transition (episode.py:84-125).
        `noise` (f32 [B, C], Exp(1)) replaces the seeded chance draw by torch's race on it; tests use it to replay the reference."""
        actions = actions.to(device=self.indices.device, dtype=torch.int32).contiguous().view(-1)
        if self._player == 0:
            self.row_actions = actions
            rewards = torch.zeros((self.batch_size,), device=self.indices.device)
        else:
            self.col_actions = actions
            self.indices, rewards = rnad_hip.transition(
                self.tree.handle(), self.indices, self.row_actions, self.col_actions, noise=noise, seed=self.seed,
                lane0=self.lane_offset, step=self._step)
            self.row_actions = None
            self.col_actions = None
        self._player = 1 - self._player
        if self._player_to_move is not None:
            self._player_to_move = 1 - self._player_to_move
        self._step += 1
        self._terminal_stale = True
        return rewards


class Episodes:
    """A parallel batch of rollout trajectories from the root (reference episode.py:131-290)."""

    _PRIMARY = ("indices", "observations", "mask_bits", "policy", "action_idx", "rewards", "values")
    _DENSE = {"mask_bits": "mask_bits", "policy": "policy", "action_idx": "actions", "rewards": "rewards"}  # attribute -> Trajectory buffer
    _observations = _values = _alive = _indices = None  # class-level defaults: objects assembled without __init__ (tests, collate) behave the same
    lane_ids = buckets = None
    _compact = None  # (rnad_hip.Trajectory(compact=True), records) of a compact bucketed rollout: dense fields expand on first access

    def __init__(self, tree: Tree, batch_size, seed=None, lane_offset=0, obs_half=False):
        self.tree: Tree = tree
        self.batch_size: int = batch_size
        self.states: States = States(tree, batch_size, seed=seed, lane_offset=lane_offset)
        self.seed = self.states.seed
        self.lane_offset = int(lane_offset)
        self.obs_half = bool(obs_half)
        self.finished: bool = False
        self.generation_time: float = 0
        self.estimation_time: float = 0
        self.t_eff: int = -1
        self.indices = self.mask_bits = self.policy = None
        self.action_idx = self.rewards = None
        self._observations = self._values = None  # lazily materialised when the rollout did not store them (bucketed rollout)
        self._compact = None
        self.lane_ids = None  # int32 [B]: lane (0-based within this batch) held by column j; None = column j is lane j
        self.buckets = None   # rnad_hip.Buckets of a bucket-ordered batch (work list of rnad_learn_bucketed)
        self.alive = None  # int32 [T + 1] on the device: lanes with indices[t] != 0
        self.actor_logits = None  # [T, B, A] raw logits of the actor (generate(keep_logits=True), native rollout only)
        self._lazy = {}

    # ---------------------------------------------------------------- lazily materialised reference attributes
    def _get(self, key, make):
        if key not in self._lazy:
            self._lazy[key] = make()
        return self._lazy[key]

    @property
    def indices(self):
        """int32 [T, B] (episode.py:218: int64 there).  A compact bucketed rollout keeps one byte per slot below the cut of the tree and
        nothing above it: the reference's tensor is rebuilt on first access (rnad_bucket_indices)."""
        value = self.__dict__.get("_indices")
        if value is None and self.__dict__.get("_compact") is not None and self.t_eff >= 0:
            value = self._compact[0].indices[: self.t_eff + 1]
            self.__dict__["_indices"] = value
        return value

    @indices.setter
    def indices(self, value):
        self.__dict__["_indices"] = value

    @property
    def observations(self):
        """[T, B, 2, A, A] (episode.py:199,221).  An observation is a function of (player to move, state) alone (episode.py:62-68):
        a rollout that did not store it (the bucketed one) gets it from K1 on first access."""
        if self._observations is None and self.indices is not None:
            T, B, A = self.t_eff + 1, self.batch_size, self.tree.max_actions
            handle = self.tree.handle()
            obs = torch.empty((T, B, 2, A, A), dtype=torch.float16 if self.obs_half else torch.float32, device=self.indices.device)
            for t in range(T):
                rnad_hip.observe(handle, self.indices[t], t & 1, obs=obs[t], half=self.obs_half)
            self._observations = obs
        return self._observations

    @observations.setter
    def observations(self, value):
        self._observations = value

    @property
    def values(self):
        """[T, B] actor values (episode.py:206,218); zeros when the rollout skipped the actor's value head (nothing reads them)."""
        if self._values is None and self.indices is not None:
            self._values = torch.zeros((self.t_eff + 1, self.batch_size), dtype=torch.float32, device=self.indices.device)
        return self._values

    @values.setter
    def values(self, value):
        self._values = value

    def _expand(self):
        """Dense mask_bits / policy / action_idx / rewards of a compact rollout (rnad_bucket_expand), on first access."""
        traj, records = self._compact
        if records is None:
            raise RuntimeError("this compact batch has no row records attached yet (Episodes.generate(compact=True, logits_table=...))")
        if traj.policy is None:
            rnad_hip.complete_records(records)  # (distinct observations: the records of the other rows of a group are copied when first read)
            rnad_hip.bucket_expand(self.tree.handle(), traj, records)
        T = self.t_eff + 1
        for name, src in self._DENSE.items():
            if self.__dict__.get("_" + name) is None:
                self.__dict__["_" + name] = getattr(traj, src)[:T]

    def invalidate_derived(self):
        """Forget everything that was derived from the rollout's own buffers (lazily built observations / one-hot actions / masks,
        the dense fields of a compact rollout): RNaD calls this after replaying the captured step, which rewrites those buffers in
        place -- the next access rebuilds from the new batch."""
        self._lazy = {}
        if self.buckets is not None:  # the bucketed rollouts do not store observations / values: both were built on access
            self._observations = None
            self._values = None
        if self._compact is not None:
            self._compact[0].invalidate()
            if self._compact[1] is not None and getattr(self._compact[1], "_expand_job", None) is not None:
                self._compact[1]._expand_stale = True  # (the replay rewrote the representatives' records: the copies are the last step's)
            self.__dict__["_indices"] = None
            self.states.indices = None
            for name in self._DENSE:
                self.__dict__["_" + name] = None

    @property
    def turns(self):
        """[T, B] int64, == t mod 2 for every lane (episode.py:197); an expanded view, no memory."""
        T = self.t_eff + 1
        return self._get("turns", lambda: (torch.arange(T, device=self.indices.device) % 2).view(T, 1).expand(T, self.batch_size))

    @property
    def actions(self):
        """[T, B, A] one-hot float of the sampled actions (episode.py:205-206)."""
        return self._get("actions", lambda: torch.nn.functional.one_hot(self.action_idx.long(), self.tree.max_actions).to(torch.float))

    @property
    def masks(self):
        """[T, B, A] legal actions of the mover == observations[:, :, 1, :, 0] (episode.py:208), expanded from `mask_bits`."""
        A = self.tree.max_actions
        return self._get("masks", lambda: ((self.mask_bits.unsqueeze(-1) >> torch.arange(A, device=self.mask_bits.device, dtype=torch.uint8)) & 1)
                         .to(torch.float))

    @property
    def q_estimates(self):
        return self._get("q_estimates", lambda: torch.zeros_like(self.policy))  # episode.py:226 (unused)

    @property
    def v_estimates(self):
        return self._get("v_estimates", lambda: torch.zeros_like(self.rewards))  # episode.py:227 (unused)

    @property
    def alive(self):
        """int32 [T + 1] on the device: lanes with indices[t] != 0 (completed first if the rollout deferred it)."""
        if self.buckets is not None and getattr(self.buckets, "alive_pending", None) is not None:
            rnad_hip.bucket_alive(self.tree.handle(), self.buckets)
        return self.__dict__.get("_alive")

    @alive.setter
    def alive(self, value):
        self.__dict__["_alive"] = value

    def norm_for_learner(self):
        """The normalisers as rnad_hip.learn_bucketed_compact wants them: when the rollout deferred its alive counts, the device buffer the
        learner's own launch is about to fill (and read, in its last kernel); valid_counts otherwise."""
        if self.buckets is not None and getattr(self.buckets, "alive_pending", None) is not None and self._compact is not None:
            return self.buckets.norm
        return self.valid_counts

    @property
    def valid_counts(self):
        """f64 [2] on the device: number of valid steps of player 0 / player 1 (= N_P of the losses)."""
        T = self.t_eff + 1
        traj = getattr(self, "_traj", None)
        if self.buckets is not None and traj is not None and T == traj.T_cap:
            if getattr(self.buckets, "alive_pending", None) is not None:  # (deferred to a learner launch that has not happened: complete it)
                rnad_hip.bucket_alive(self.tree.handle(), self.buckets)
            return self.buckets.norm  # counted by the rollout itself (rnad_rollout_bucketed); callers must not modify it in place
        a = self.alive[:T].to(torch.float64)
        return torch.stack([a[0::2].sum(), a[1::2].sum()])

    # ---------------------------------------------------------------- episode.py:175-230
    def generate(self, net: torch.nn.Module, noise_action=None, noise_chance=None, max_steps=None, trim=True, keep_logits=False,
                 skip_absorbed=False, store_values=True, tabular=None, bucketed=False, logits_table=None, value_table=None, policy_table=None, step_params=None,
                 compact=False, visited=None, defer_alive=False, staged_actor=None, learn=None):
        """Play the batch to the end with `net` as the actor.

        Nets exposing `forward_logits(obs) -> (logits [B,A], value [B,1])` (the MLP here) take the fast path: policy head,
        sampling, transition and the next observation are HIP kernels.  Any other module honouring the reference
        contract `forward(obs) -> (logits, policy, value, actions)` (nn/net.py:37-51) is driven through that instead and
        samples for itself.  noise_action [T,B,A] / noise_chance [T,B,C]: explicit Exp(1) noise (tests).

        trim=True reads the alive counters back (the rollout's only host sync) and cuts the trajectory to the reference's
        length T (episode.py:194 stops once every lane is absorbed).  trim=False keeps all T_cap = 2 * depth steps and
        never synchronises: trailing steps where every lane sits in state 0 are invalid (`indices == 0`) and contribute
        nothing to V-trace or the losses, so learning is unchanged; RNaD uses this.

        skip_absorbed=True (native MLP actor on a ragged tree only): from step 1 on the actor is evaluated on the lanes that
        are still in the tree; an absorbed lane keeps the logits / value of its last live step, where the reference stores the
        net's output on state 0's observation (episode.py:203-212).  Those slots are invalid (`indices == 0`) for every
        consumer, so RNaD uses this too; leave it off to reproduce the reference's buffers slot for slot.

        tabular (native MLP actor only; default: whenever the tree is small next to the batch, 8 S <= T B): the actor is evaluated
        once on the 2S distinct observations of the tree (an observation depends on the state and the player to move only,
        episode.py:62-68) and each step gathers its logits / value row -- the same rollout bit for bit (same inputs, same
        kernel) at 2S instead of T*B net evaluations.  tabular=False evaluates the net on every lane at every step.

        bucketed=True (tabular actor only; what RNaD's per-row update uses): the same episodes, but column j of every buffer holds
        lane `lane_ids[j]` -- the lanes stably sorted by the state they reach at a fixed depth of the tree -- and `buckets` carries
        the work list of rnad_learn_bucketed; `observations` are not stored (materialised on first access).  Falls back to the
        lane-ordered rollout when the tree cannot be bucketed.  logits_table / value_table: the actor already evaluated on the
        tree's 2S observations (row = player * S + state; logits in the first A columns), instead of evaluating `net` here.
        policy_table = (table, column): additionally the actor's POLICY rows (A floats from `column` on, e.g. inside
        rnad_hip.bucket_records), which the bucketed rollout reads instead of taking the policy head of the logits itself.
        step_params (bucketed only): a device int64 [2] holding struct rnad_step_params; the kernels then take the noise seed from
        there instead of `self.seed` (a captured graph of the step is replayed with a new seed).

        compact=True (bucketed with policy_table = (rnad_hip.bucket_records(...), rnad_hip.policy_column(A)), store_values=False,
        at most 21 steps): the rollout stores 64 bytes per lane -- `indices`, the packed actions and the episode's one non-zero
        reward (rnad_rollout_bucketed_compact) -- which is all RNaD's on-policy update reads; `mask_bits`, `policy`, `action_idx`,
        `rewards` (and what derives from them) are written by rnad_bucket_expand on first access.  Slots of absorbed lanes then show
        action 0 where the dense rollout keeps drawing (nothing reads them).  With logits_table instead of policy_table the actor is
        the policy head of those logits and the caller attaches the records later.  visited (int32 [2S], compact only): receives a
        1 for every (player, state) row a live slot of the batch sits in.  defer_alive (compact, trim=False): the per-step alive
        counters and the loss normalisers are added up by the learner's launch (rnad_hip.learn_bucketed_compact) instead of by a
        kernel of their own; reading `alive` or `valid_counts` before that completes them on the spot.  learn (compact with policy_table,
        trim=False; a dict: fast_records, hp, norm_is_global, rows, groups, leaf): the batch's on-policy update is added up by the very launch
        that plays it (rnad_hip.rollout_learn_bucketed_compact) and `_learned` carries its per-row gradient tables.  staged_actor (compact with
        logits_table; trees that are large next to the batch): a callable `f(rows)` that evaluates the actor's logits INTO
        logits_table on the given rnad_hip.LiveRows / RowList -- called twice: with the rows of the cut's upper states before the keys
        pass and the sort, then with the rows of the groups the batch turned out to descend into, before the rollout itself
        (rnad_bucket_sort / rnad_bucket_play).

        store_values=False (native MLP actor only): the actor's value head is not evaluated and `values` is zeros.  The
        reference stores the actor's values (episode.py:206,218) but nothing ever reads them (learn/rnad.py:373 recomputes v
        with the learner net), so RNaD's own rollouts do without, unless `reuse_actor_outputs` needs them.
        """
        tree, B = self.tree, self.batch_size
        handle = tree.handle()
        T_cap = 2 * handle.max_depth if max_steps is None else int(max_steps)
        dev = torch.device(tree.device)
        fast = hasattr(net, "forward_logits")
        # weights are fixed for the whole rollout: pack them once -- unless the caller hands the actor's table in (nothing to evaluate)
        packed = net.pack() if fast and hasattr(net, "pack") and logits_table is None and policy_table is None else None
        net.eval()
        time_start = time.perf_counter()
        native = ((packed is not None or logits_table is not None or policy_table is not None) and noise_action is None
                  and noise_chance is None and type(net).forward_logits is _MLP.forward_logits)
        if tabular is None:
            tabular = 8 * handle.S <= T_cap * B
        tabular = native and tabular and not keep_logits
        bucketed = bool(bucketed) and tabular and T_cap <= 64 and rnad_hip.bucket_plan(handle, B) is not None
        # compact: the actor is the pi columns of a records table (the learner then reads them as the acting policy), or -- records not
        # made yet (RNaD's lazy rows) -- the logits table they will be made from
        compact = (bool(compact) and bucketed and not store_values and T_cap <= rnad_hip.COMPACT_MAX_STEPS
                   and (policy_table[1] == rnad_hip.policy_column(tree.max_actions) if policy_table is not None else logits_table is not None))
        traj = rnad_hip.Trajectory(handle, B, T_cap, dev, half=self.obs_half, with_observations=not bucketed,
                                   with_values=store_values or not bucketed, compact=compact)
        self.buckets = self.lane_ids = self._compact = None
        if tabular:
            # one actor evaluation per (player, state), then the whole loop natively with per-lane gathers
            table, vtable = logits_table, value_table
            if table is None and not (bucketed and policy_table is not None):
                packed = packed if packed is not None else net.pack()
                table, vtable = rnad_hip.mlp_forward(packed, net.width, handle.observations_table(self.obs_half), tree.max_actions,
                                                     want_value=store_values)
            defer_alive = bool(defer_alive) and compact and not trim
            self.__dict__.pop("_learned", None)
            if compact and policy_table is not None and learn is not None and visited is None and not trim:
                # rollout AND the on-policy update of the batch in one launch (rnad_rollout_learn_bucketed_compact): RNaD.__learn picks
                # the per-row gradient tables up from `_learned` instead of launching the learner
                self.buckets, dlogit, dv = rnad_hip.rollout_learn_bucketed_compact(
                    handle, traj, policy_table[0], learn["fast_records"], learn["hp"], seed=self.seed, lane0=self.lane_offset,
                    step_params=step_params, norm_is_global=learn.get("norm_is_global", True), rows=learn.get("rows"), groups=learn.get("groups"),
                    distinct=bool(learn.get("distinct", False)), norm_global=learn.get("norm_global"), leaf=learn.get("leaf"))
                self._learned = dict(records=policy_table[0], dlogit=dlogit, dv=dv)
                self.lane_ids = self.buckets.lane_ids
                self._compact = (traj, policy_table[0])
            elif compact and policy_table is not None:
                self.buckets = rnad_hip.rollout_bucketed_compact(handle, traj, policy_table[0], seed=self.seed, lane0=self.lane_offset,
                                                                 step_params=step_params, visited=visited, defer_alive=defer_alive)
                self.lane_ids = self.buckets.lane_ids
                self._compact = (traj, policy_table[0])
            elif compact and staged_actor is not None:
                upper = rnad_hip.bucket_upper_rows(handle, B)
                staged_actor(upper)
                # an actor that also leaves its POLICY rows (rnad_hip.mlp_forward_actor: logits_table._policy_rows) spares the two
                # policy-head launches the sort and the rollout would otherwise take from the logits
                pol = getattr(table, "_policy_rows", None)
                actor = dict(table=pol, table_is_policy=True) if pol is not None else dict(table=table)
                stage = rnad_hip.bucket_stage(handle, B) if (pol is not None and os.environ.get("RNAD_STAGE_LEVELS", "2") != "1") else None
                if stage is not None:
                    # two more stages below the cut (rnad_bucket_stage_*): the roots of the group subtrees the lanes enter, then -- one drawn
                    # transition later -- the subtrees below the states they land in (configs[3]: 176 k rows instead of the 807 k of
                    # every non-empty group)
                    self.buckets, _, _ = rnad_hip.bucket_sort(handle, traj, seed=self.seed, lane0=self.lane_offset, step_params=step_params,
                                                              visited=visited, want_rows=False, stage=stage, **actor)
                    roots = rnad_hip.bucket_stage_rows(handle, B, stage, 0, seed=self.seed, step_params=step_params)
                    staged_actor(roots)
                    rnad_hip.bucket_stage_walk(handle, traj, self.buckets, stage, pol, seed=self.seed, lane0=self.lane_offset,
                                               step_params=step_params)
                    rows = rnad_hip.bucket_stage_rows(handle, B, stage, 1, seed=self.seed, step_params=step_params)
                    staged_actor(rows)
                    self.staged_rows = (upper, roots, rows)  # (bench.py reads how many rows the actor was evaluated on)
                else:
                    # (the sort's last kernel writes the list of rows still to evaluate and clears `visited`: no flags, no compaction)
                    self.buckets, rows, _ = rnad_hip.bucket_sort(handle, traj, seed=self.seed, lane0=self.lane_offset, step_params=step_params,
                                                                 visited=visited, **actor)
                    staged_actor(rows)
                    self.staged_rows = (upper, rows)  # (bench.py reads how many rows the actor was evaluated on)
                rnad_hip.bucket_play(handle, traj, self.buckets, rows=None if pol is not None else rows, seed=self.seed, lane0=self.lane_offset,
                                     step_params=step_params, visited=visited, defer_alive=defer_alive, visited_is_clear=visited is not None, **actor)
                self.lane_ids = self.buckets.lane_ids
                self._compact = (traj, None)  # the caller attaches the records once they exist (learn/rnad.py, lazy rows)
            elif compact:
                self.buckets = rnad_hip.rollout_bucketed_compact(handle, traj, table, seed=self.seed, lane0=self.lane_offset,
                                                                 step_params=step_params, table_is_policy=False, visited=visited,
                                                                 defer_alive=defer_alive)
                self.lane_ids = self.buckets.lane_ids
                self._compact = (traj, None)  # the caller attaches the records once they exist (learn/rnad.py, lazy rows)
            elif bucketed and policy_table is not None:
                self.buckets = rnad_hip.rollout_bucketed(handle, traj, policy_table[0], vtable if store_values else None, seed=self.seed,
                                                         lane0=self.lane_offset, table_is_policy=True, column=policy_table[1],
                                                         step_params=step_params)
                self.lane_ids = self.buckets.lane_ids
            elif bucketed:
                self.buckets = rnad_hip.rollout_bucketed(handle, traj, table, vtable if store_values else None, seed=self.seed,
                                                         lane0=self.lane_offset, step_params=step_params)
                self.lane_ids = self.buckets.lane_ids
            else:
                rnad_hip.rollout_run_tabular(handle, traj, table[:, : tree.max_actions].contiguous(), vtable if store_values else None,
                                             seed=self.seed, lane0=self.lane_offset)
        elif native:
            # the actor is this package's MLP: the whole loop is enqueued natively (rnad_rollout_run)
            packed = packed if packed is not None else net.pack()
            self.actor_logits = rnad_hip.rollout_run(handle, traj, net.width, packed, seed=self.seed, lane0=self.lane_offset,
                                                     keep_logits=keep_logits,
                                                     skip_absorbed=skip_absorbed and not keep_logits and not handle.uniform_length,
                                                     store_values=store_values)
        else:
            rnad_hip.rollout_begin(handle, traj)
            with torch.no_grad():
                for t in range(T_cap):
                    obs_t = traj.observations[t]
                    na = None if noise_action is None else noise_action[t]
                    nc = None if noise_chance is None else noise_chance[t]
                    if fast:
                        logits, value = net.forward_logits(obs_t, packed=packed) if packed is not None else net.forward_logits(obs_t)
                        rnad_hip.rollout_step(handle, traj, t, value.reshape(-1), logits=logits, noise_action=na, noise_chance=nc,
                                              seed=self.seed, lane0=self.lane_offset)
                    else:
                        _, policy, value, actions = net.forward(obs_t)
                        rnad_hip.rollout_step(handle, traj, t, value.reshape(-1).contiguous(), policy=policy.contiguous(),
                                              actions=actions.to(torch.int32).contiguous().view(-1), noise_chance=nc,
                                              seed=self.seed, lane0=self.lane_offset)
            rnad_hip.rollout_end(handle, traj)
        if trim:
            alive = traj.alive.cpu()  # the only host sync of the rollout
            T = int((alive[:T_cap] > 0).sum().item())
        else:
            T = T_cap
        time_end = time.perf_counter()
        self.generation_time = time_end - time_start
        self._traj = traj
        self.t_eff = T - 1
        self.indices = None if compact else traj.indices[:T]  # (compact: rebuilt from the relative states on first access)
        self.observations = traj.observations[:T] if traj.observations is not None else None
        for name, src in self._DENSE.items():
            setattr(self, name, None if compact else getattr(traj, src)[:T])
        self.values = traj.values[:T] if traj.values is not None else None
        self.alive = traj.alive[: T + 1]
        if self.actor_logits is not None:
            self.actor_logits = self.actor_logits[:T]
        self._lazy = {}
        self.states.indices = None if compact else traj.indices[T]
        self.states._final_of = (traj, T) if compact else None  # (compact: States.indices reads row T of the rebuilt tensor on first access)
        self.states.terminal = True  # by construction after 2 * depth steps
        self.finished = True
        net.train()

    def __repr__(self):
        result = ""
        for key in ("batch_size", "t_eff", "finished", "generation_time") + self._PRIMARY:
            value = getattr(self, key)
            if torch.is_tensor(value) and torch.numel(value) > 20:
                value = value.shape
            result += f"{key}: {value}\n"
        return result

    # ---------------------------------------------------------------- episode.py:243-256
    def _like(self, batch_size):
        result = Episodes(self.tree, batch_size, seed=self.seed, lane_offset=self.lane_offset, obs_half=self.obs_half)
        result.finished = True
        return result

    def sample(self, batch_size, shuffle=False, selected=None):
        """A uniformly random subset of `batch_size` lanes (reference: `random.sample` + index_select on every tensor).
        selected: take exactly these lanes, in this order (an int64 index tensor / sequence) instead of drawing them -- how the
        tests replay the reference's own `random.sample` draws.

        Asking for the whole batch -- what RNaD does every step with the default one-batch buffer (rnad.py:507) -- is a
        pure permutation of lanes in the reference, and no loss term depends on the lane order, so it returns `self`
        without copying 1.7 GB per step; pass shuffle=True to get the permuted copy anyway.  Proper subsets are drawn with
        a device permutation seeded from python's `random`."""
        assert self.finished
        batch_size = min(batch_size, self.batch_size)
        dev = torch.device(self.tree.device)  # (not self.indices.device: a compact batch would rebuild its indices for it)
        drawn = selected is None
        if selected is not None:
            selected = torch.as_tensor(selected, dtype=torch.long, device=dev)
            assert selected.numel() == batch_size, "`selected` must list min(batch_size, self.batch_size) lanes"
        elif batch_size == self.batch_size and not shuffle:
            return self
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(random.getrandbits(62))
            selected = torch.randperm(self.batch_size, generator=g, device=dev)[:batch_size]
        result = self._like(batch_size)
        for key in self._PRIMARY:
            setattr(result, key, torch.index_select(getattr(self, key), dim=1, index=selected))
        result.t_eff = self.t_eff
        if self.lane_ids is not None:
            result.lane_ids = torch.index_select(self.lane_ids, 0, selected)
        if drawn and batch_size == self.batch_size:
            result.alive = self.alive  # a permutation keeps the per-step counts
        else:
            alive = torch.zeros_like(self.alive)
            alive[: self.t_eff + 1] = (result.indices != 0).sum(dim=1).to(torch.int32)
            result.alive = alive
        return result

    # ---------------------------------------------------------------- episode.py:258-290
    @classmethod
    def collate(cls, lst: list["Episodes"]):
        """Pad each member along time with zeros (index 0 == invalid) and concatenate along the batch."""
        if len(lst) == 1:
            return lst[0]  # nothing to pad or concatenate
        t_eff = max(e.t_eff for e in lst)
        tree = lst[0].tree
        batch_size = sum(e.batch_size for e in lst)
        assert all(e.tree == tree for e in lst)
        assert all(e.finished for e in lst)
        result = lst[0]._like(batch_size)
        T = t_eff + 1
        for key in cls._PRIMARY:
            parts = []
            for e in lst:
                x = getattr(e, key)
                if x.shape[0] < T:
                    pad = torch.zeros((T - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                    x = torch.cat([x, pad], dim=0)
                parts.append(x)
            setattr(result, key, parts[0] if len(parts) == 1 else torch.cat(parts, dim=1))
        alive = torch.zeros((T + 1,), dtype=torch.int32, device=result.indices.device)
        alive[:T] = (result.indices != 0).sum(dim=1).to(torch.int32)
        result.alive = alive
        result.t_eff = t_eff
        return result


def _dense_field(name):
    """mask_bits / policy / action_idx / rewards: plain attributes, except that a compact rollout fills them on first access."""
    private = "_" + name

    def get(self):
        value = self.__dict__.get(private)
        if value is None and self.__dict__.get("_compact") is not None:
            self._expand()
            value = self.__dict__.get(private)
        return value

    def put(self, value):
        self.__dict__[private] = value

    return property(get, put)


for _name in Episodes._DENSE:
    setattr(Episodes, _name, _dense_field(_name))


class Buffer:
    """Replay buffer of Episodes played with older actor nets (reference episode.py:292-333)."""

    def __init__(self, max_size) -> None:
        self.max_size = max_size
        self.episodes_buffer = deque(maxlen=max_size)

    def sample(self, batch_size):
        n = len(self.episodes_buffer)
        bucket_sizes = numpy.random.multinomial(batch_size, [1 / n] * n)
        assert sum(bucket_sizes) == batch_size
        return Episodes.collate([self.episodes_buffer[_].sample(int(bucket_sizes[_])) for _ in range(n)])

    def append(self, episodes: Episodes):
        self.episodes_buffer.append(episodes)
        while len(self.episodes_buffer) > self.max_size:
            self.episodes_buffer.popleft()

    def clear(self):
        self.episodes_buffer.clear()
