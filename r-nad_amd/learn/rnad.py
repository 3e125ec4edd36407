"""RNaD trainer -- drop-in for reference learn/rnad.py (same constructor kwargs, `run()`, schedule and checkpoint formats).

The host loop is the reference's (`__resume`, learn/rnad.py:458-531): (m, n) schedule, alpha ramp, rollout cadence, Adam,
EMA target, regularisation-net rotation.  What one iteration executes depends on `RNaD.tabular`:

  True (default, trees that are small next to the batch)
            the nets on the tree's 2S observations (fused fp32-MFMA kernels, one launch) -> row records -> bucket-ordered rollout
            keeping 64 bytes per lane (compact_trajectory) -> rnad_learn_bucketed_compact (V-trace / NeuRD per lane from
            row-precomputed operands, gradients summed per (player, state) row in LDS) -> one MLP backward over the 2S rows ->
            clip + Adam + EMA in one launch (fused_optimizer); the whole step is captured once as a hipGraph and replayed
            (train_step).
  "forward" nets on the 2S observations, per-slot gradients, per-slot MLP backward: bit-identical to the dense mode.
  False     rollout with K1/K2/K3 around the fused MLP forward on every lane; four MLP forwards over the trajectory -> ONE fused
            kernel (policy heads, process_policy, both players' V-trace, NeuRD + value loss gradients, learn/rnad.py:365-425) ->
            fused MLP backward with the closed-form dL/dlogit, dL/dv -> clip -> Adam -> EMA.  What the reference does, slot for slot.

Data parallel: when torch.distributed is initialised (one process per GPU, backend "nccl" == RCCL over xGMI), every rank
plays `batch_size // world_size` lanes of the SAME seeded draws (global lane ids), the two loss normalisers are
all-reduced (they are batch-global, learn/vtrace.py:373,388; beside the learner kernel, which sums un-normalised addends), and the
10 756 parameter gradients are all-reduced (sum) in one flat bucket before clipping.  No other communication.
"""
import logging
import os
import time
from typing import Dict

import torch
import torch.nn as nn
import torch.distributed as dist

import environment.episode as episode
import environment.tree as tree
import nn.net as net
import util.metric as metric
import rnad_hip
from learn import checkpoint


def _save_root():
    return os.environ.get("RNAD_SAVE_DIR") or os.path.join(os.path.dirname(os.path.realpath(__file__)), "..")


def _dist_on():
    """Data parallel whenever a process group exists -- also a one-rank group: the collectives then still run (through RCCL on
    a GPU), which is what lets a single-GPU box exercise the exact call sequence of the N-rank update."""
    return dist.is_available() and dist.is_initialized()


class RNaD:
    def __init__(
        self,
        tree: tree.Tree,
        device=torch.device("cuda"),
        directory_name=None,
        batch_size=3 * 2**8,
        eta=0.2,
        bounds=[100, 165, 200],
        delta_m=[10_000, 100_000, 35_000],
        lr=5 * 10**-5,
        logit_clip=2,
        neurd_clip=10**3,
        grad_clip=10**3,
        b1_adam=0,
        b2_adam=0.999,
        epsilon_adam=10**-8,
        gamma_averaging=0.001,
        roh_bar=1,
        c_bar=1,
        epsilon_threshold=0.03,
        n_discrete=32,
        # parameters from https://arxiv.org/abs/2206.15378 (reference learn/rnad.py:40-64)
        n_batches_per_buffer=1,
        buffer_mod=1,
        net_params=None,
        vtrace_gamma=1,
        value_loss_weight=1,
        neurd_loss_weight=1,
        wandb=False,
        use_same_init_net_as=False,
    ):
        self.tree = tree
        self.tree_hash = 0
        self.device = device

        self.eta = eta
        self.bounds = bounds
        self.delta_m = delta_m
        self.n_batches_per_buffer = n_batches_per_buffer
        self.buffer_mod = buffer_mod
        self.lr = lr
        self.beta = logit_clip
        self.neurd_clip = neurd_clip
        self.grad_clip = grad_clip
        self.b1_adam = b1_adam
        self.b2_adam = b2_adam
        self.epsilon_adam = epsilon_adam
        self.gamma_averaging = gamma_averaging
        self.roh_bar = roh_bar
        self.c_bar = c_bar
        self.batch_size = batch_size
        self.epsilon_threshold = epsilon_threshold
        self.n_discrete = n_discrete
        self.vtrace_gamma = vtrace_gamma
        self.neurd_weight = neurd_loss_weight
        self.value_weight = value_loss_weight
        self.wandb = wandb

        if directory_name is None:
            directory_name = str(int(time.perf_counter()))
        self.directory_name = directory_name

        if net_params is None:
            net_params = {"type": "MLP", "max_actions": self.tree.max_actions, "width": 2**8}
        self.net_params = net_params

        self.saved_keys = [key for key in self.__dict__.keys() if key != "tree"]
        # only the above members are saved in and reloaded from the 'params' object (reference learn/rnad.py:153)

        self.directory = os.path.join(_save_root(), "saved_runs", directory_name)
        self.use_same_init_net_as = use_same_init_net_as

        self.m = 0
        self.n = 0
        self.total_steps = 0
        self.net = None
        self.net_target = None
        self.net_reg = None
        self.net_reg_ = None
        self.last_log = None  # scalars of the most recent logged step (the reference sends them to wandb)
        self.keep_last_log = False  # True: compute them every log_mod steps even without wandb (about ten host syncs per logged step)
        # On-policy shortcut (off by default): with the default one-batch buffer the learner net of __learn IS the actor of the
        # rollout that just finished, with unchanged weights, so forward_batch(net) (rnad.py:373) recomputes bit-identical
        # logits / values; when True they are taken from the rollout and only the backward runs.
        self.reuse_actor_outputs = False
        # Tabular net evaluation: observations are a function of (state, player to move), so on a tree that is small next to the
        # batch (tabular_gate * S <= T * B) the nets are evaluated on the 2S distinct observations instead of on every (t, b) slot
        # -- see __learn.  Rollouts, per-slot net outputs, V-trace targets and losses are the dense path's bits in every mode.
        #   True (default)  the per-slot gradients are also summed per (player, state) row (64-bit fixed point: reproducible bit
        #                   for bit) and ONE backward over the 2S observations gives the weight gradients -- equal to the dense
        #                   path's up to fp32 summation order (1e-5 of the largest entry; pinned against the reference's own
        #                   gradients in tests/test_hip_parity.py and tests/test_hip_replay.py);
        #   "forward"       only the forward evaluations are deduplicated, the backward runs per slot: weight gradients
        #                   bit-identical to the dense path;
        #   False           every net on every slot, as the reference does.
        self.tabular = True
        self.tabular_gate = 8
        self.use_graph = True  # capture the on-policy tabular step as a hipGraph and replay it (train_step)
        # the on-policy step keeps 64 bytes per lane of its batch (states, packed actions, one reward): csrc/bucket.hip COMPACT;
        # the dense Episodes fields are written when something reads them
        self.compact_trajectory = True
        # trees that are large next to the batch (None: when 2S > lanes per rank): only the learner's policy head runs on all 2S rows;
        # the value heads, the row records, the gradient tables and the backward cover the rows the batch visited (5 % on configs[3])
        self.lazy_rows = None
        self.last_rows = None  # rnad_hip.LiveRows of the last lazy step
        self.fused_optimizer = True  # clip + Adam + EMA target of the MLP in one launch (csrc/optim.hip) instead of ~8 torch launches
        # Data parallel, per-row mode: what a rank does on the 2S rows of the tree (table forwards + records, backward) does not shrink
        # with its share of the batch.  shard_rows = True also shards THAT: rank r evaluates rows [r * 2S / N, (r + 1) * 2S / N), the
        # record tables are all-gathered, the learner's 64-bit per-row sums are all-reduced (exact: every rank then holds the sums of
        # the GLOBAL batch, bit for bit those of a one-process step), and a rank finishes and back-propagates its own rows only; the
        # 43 KB weight-gradient all-reduce closes the step as before.  Default off: three more collectives per step, to be set against the
        # row work they remove on real xGMI (DESIGN.md section 7 has the byte counts and the prediction).
        self.shard_rows = False
        # Distinct observations (csrc/rows_dedup.hip): rows of the tree with the same observation share their net outputs and records, and
        # their gradients are added up before the backward -- the table launch and the backward then run on one representative row per
        # observation (BASELINE configs[1]: 132 862 rows, ~15 k distinct observations, because the deepest level's payoff matrices are
        # +-1).  Exact for the forward (same bits per row); the weight gradient is the same sum in another fp32 order.  Applied when it
        # removes at least a fifth of the rows; never on logging steps, lazy rows or row sharding.
        self.dedup_rows = True
        # Leaf paths (_leaf_now): the learner of the one-call rollout + learner on the tree's terminal transitions, weighted with the lanes
        # that left the tree by them, instead of once per lane.  None: automatic (uniform-length trees with no more leaf paths than lanes).
        self.leaf_paths = None
        # alpha_ahead(k): the alpha the caller will pass k steps from now (run() sets it from rnad.py:497's schedule); None: "as now".
        # Only a prediction -- a replayed step whose scalars were not the queued ones sets them itself (_graph_step).
        self.alpha_ahead = None
        # ... added up by k_bucket_finish itself (the `groups` of rnad_bucket_finish) when no row above the cut shares its observation;
        # False: always rnad_rows_segment_sum on the tables of a finish over all rows (the same bits, one launch more).
        self.group_sums_in_finish = True
        # The legal fold (include/rnad_hip.h): on a tree whose observation rows all carry the same legal plane (all ones; e0 in the
        # absorbing state) the table evaluations of the per-row mode run the MLP with A^2 + 1 input features instead of 2 A^2 -- the
        # same function of the same weights in another summation order, ~45 % fewer matrix instructions in the first layer.
        self.fold_legal = True
        # ragged trajectories: evaluate / differentiate the nets on live (t, b) slots only (see __learn); same losses and gradients
        self.skip_absorbed = True
        self.obs_half = False  # store observations as fp16 (BASELINE.json configs[4]); arithmetic stays fp32
        self.nashconv_history = []  # (m, total_steps, nashconv)

    # ------------------------------------------------------------------ data-parallel helpers
    def _dp(self):
        """Data parallel over the default process group -- unless this trainer was told to stand alone (data_parallel = False:
        bench.py's single-GPU reference leg inside an N-rank run)."""
        return getattr(self, "data_parallel", True) and _dist_on()

    @property
    def _rank(self):
        return dist.get_rank() if self._dp() else 0

    @property
    def _world(self):
        return dist.get_world_size() if self._dp() else 1

    def _plays_what_it_learns(self, handle, local_batch, T_cap, buffer):
        """This step plays its batch with the compact bucketed rollout and learns from exactly that batch, in bucket order: a one-batch
        buffer refilled every step (Buffer.sample then hands the batch back in place, buckets and all).  ONE predicate for everything
        that relies on it -- the record tables written for a subset of the rows only (distinct observations, row sharding: their logit /
        v / v_target tables are valid in the listed rows alone, which only the bucketed learner never reads), the copies of the distinct
        observations' records carried by the rollout's keys pass, lazy rows, the one-launch rollout + learner.  Episodes.generate's own
        `compact` condition is the same list (environment/episode.py)."""
        return bool(self.buffer_mod == 1 and getattr(buffer, "max_size", None) == 1 and getattr(self, "compact_trajectory", True)
                    and T_cap <= rnad_hip.COMPACT_MAX_STEPS and not self.reuse_actor_outputs
                    and not getattr(self, "store_actor_values", False) and rnad_hip.bucket_plan(handle, local_batch) is not None)

    def _shard_now(self, handle, local_batch, log, lazy, on_policy=True):
        """Row sharding applies: asked for, more than one rank, the default per-row step with the fused table launch, not a logging step,
        the batch learned from in bucket order (_plays_what_it_learns), and the GLOBAL batch within the headroom of the 64-bit per-row sums
        (csrc/bucket.hip kLaneBits: one addend per lane of up to 2^22 lanes -- the all-reduce adds every rank's lanes into one sum, so
        with this bound the reduced sums cannot wrap).  The per-addend overflow flag stays per rank and is not reduced: a rank that saw an
        addend beyond the fixed-point range poisons ITS gradient tables with NaN (k_bucket_finish), its weight gradients are NaN, and the
        43 KB gradient all-reduce that closes the step hands the NaN to every rank -- the failure is loud everywhere one collective later."""
        A = self.tree.max_actions
        return bool(getattr(self, "shard_rows", False) and self._dp() and self._world > 1 and log is None and not lazy and on_policy
                    and self.batch_size <= rnad_hip.BUCKET_MAX_LANES
                    and rnad_hip.mlp_rows_records_supported(A, self.net.width, self._fold())
                    and rnad_hip.bucket_plan(handle, local_batch) is not None)

    def _dedup_now(self, handle, log, lazy, shard, fold, on_policy=True):
        """TreeHandle.obs_dedup() when the step should evaluate the nets on distinct observations only, else None.  on_policy:
        _plays_what_it_learns -- any other step (a replay buffer of several batches, stored actor values) reads the per-row logit / value
        tables, which the launch on the representatives leaves unwritten in every other row."""
        if not getattr(self, "dedup_rows", True) or log is not None or lazy or shard or not on_policy:
            return None
        if not rnad_hip.mlp_rows_records_supported(self.tree.max_actions, self.net.width, fold):
            return None
        d = handle.obs_dedup(getattr(self, "obs_half", False))
        return d if 5 * d.n_unique <= 4 * d.n_rows else None

    def _row_shard(self, handle):
        """(RowList of this rank's rows, rows per rank): rank r owns rows [r * per, min((r + 1) * per, 2S)) of the (player, state) tables."""
        world, rank = self._world, self._rank
        key = (id(handle), world, rank)
        cached = self.__dict__.get("_row_shard_cache")
        if cached is None or cached[0] != key:
            N = 2 * handle.S
            per = (N + world - 1) // world
            r0, r1 = min(N, rank * per), min(N, (rank + 1) * per)
            cached = self._row_shard_cache = (key, rnad_hip.RowList(torch.arange(r0, r1, dtype=torch.int32), N, self.device), per)
        return cached[1], cached[2]

    def _gather_rows(self, t, per):
        """t [world * per, k]: every rank wrote its own `per` rows; afterwards every rank holds them all."""
        rank, world = self._rank, self._world
        mine = t[rank * per: (rank + 1) * per]
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(t, mine)  # in place: the input is this rank's slice of the output
        else:
            dist.all_gather([t[i * per: (i + 1) * per] for i in range(world)], mine.clone())

    def _sync_from_rank0(self, module):
        if self._dp():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0)

    def _new_seed(self):
        """Noise seed of the next rollout.  The first one is drawn from torch's generator (so torch.manual_seed makes runs
        repeatable) and, under torch.distributed, broadcast from rank 0 -- once: the following seeds are a counter hashed with
        it on the host, so the ranks stay in lock step without a broadcast + host sync in every step."""
        if getattr(self, "_seed_override", None) is not None:
            return self._seed_override
        if getattr(self, "_seed_base", None) is None:
            s = torch.randint(0, 2**62, (1,), dtype=torch.int64)
            if self._dp():
                s = s.to(self.device)
                dist.broadcast(s, src=0)
            self._seed_base, self._seed_count = int(s.item()), 0
            return self._seed_base
        self._seed_count += 1
        return self._seed_at(self._seed_count)

    @staticmethod
    def alpha_of(n, delta_m):
        """alpha of step n of an outer iteration of delta_m steps (rnad.py:497)."""
        return 1 if n > delta_m / 2 else n * 2 / delta_m

    def _seed_at(self, count):
        """The count-th seed after the first (count >= 1): what _new_seed returns on its (count + 1)-th call."""
        z = (self._seed_base + count * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF  # splitmix64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return (z ^ (z >> 31)) & (2**62 - 1)

    # ------------------------------------------------------------------ reference learn/rnad.py:174-188
    def __new_net(self) -> nn.Module:
        types = {"MLP": net.MLP}
        if hasattr(net, "ConvNet"):
            types["ConvNet"] = net.ConvNet
        t = types[self.net_params["type"]]
        net_params = {k: v for k, v in self.net_params.items() if k != "type"}
        net_params["device"] = self.device
        new_net = t(**net_params)
        new_net.eval()
        return new_net

    def __new_optimizer(self):
        # fused=True: one kernel for all eight tensors instead of ~10 foreach launches (same update rule)
        # capturable=True: the step counter lives on the device, so optimizer.step() can be part of a captured graph (train_step)
        on_gpu = isinstance(self.device, torch.device) and self.device.type == "cuda"
        return torch.optim.Adam(self.net.parameters(), lr=self.lr, betas=(float(self.b1_adam), float(self.b2_adam)),
                                eps=self.epsilon_adam, fused=on_gpu, capturable=on_gpu)

    # ------------------------------------------------------------------ run directory: fresh start or resume
    # Behaviour of reference learn/rnad.py:190-319 (a run directory with checkpoints is resumed from its last one, anything else
    # starts fresh and writes `params` + checkpoint 0/0), rebuilt around learn/checkpoint.RunStore.  Under torch.distributed the
    # ranks share the directory: rank 0 alone looks at the disk and decides, the others receive the decision, and every read of a
    # file rank 0 writes sits behind a barrier.
    def _state_nets(self):
        return {"net": self.net, "net_target": self.net_target, "net_reg": self.net_reg, "net_reg_": self.net_reg_}

    def _barrier(self):
        if self._dp():
            dist.barrier()

    def __initialize(self):
        logging.info("R-NaD run '%s' in %s", self.directory_name, self.directory)
        self._store = store = checkpoint.RunStore(self.directory)
        decision = [store.latest() if self._rank == 0 else None]
        if self._dp():
            dist.broadcast_object_list(decision, src=0)
        resume_from = decision[0]
        if resume_from is None:
            self._start_fresh()
        else:
            self._resume_from(*resume_from)
        self._barrier()  # rank 0 has finished writing before anyone returns (and possibly re-reads the directory)
        if self.wandb:
            import wandb

            wandb.init(resume=resume_from is not None, project="RNaD", config={key: self.__dict__[key] for key in self.saved_keys})
            wandb.run.name = self.directory_name

    def _start_fresh(self):
        self.tree_hash = self.tree.hash
        self.m = self.n = 0
        self.net = self.__new_net()
        if self.use_same_init_net_as:  # share the initial weights of another run: its checkpoint 0/0 (rnad.py:213-224)
            other = checkpoint.RunStore(os.path.join(_save_root(), "saved_runs", self.use_same_init_net_as))
            self.net.load_state_dict(other.read(0, 0, map_location=self.device)["net"])
            logging.info("initial net taken from run '%s'", self.use_same_init_net_as)
        self._sync_from_rank0(self.net)
        self.net.train()
        for name in ("net_target", "net_reg", "net_reg_"):  # all four nets start equal (rnad.py:226-231)
            clone = self.__new_net()
            clone.load_state_dict(self.net.state_dict())
            setattr(self, name, clone)
        self.optimizer = self.__new_optimizer()
        if self._rank == 0:
            self._store.write_params({key: self.__dict__[key] for key in self.saved_keys})
        self.__save_checkpoint()

    def _resume_from(self, m, n):
        params = self._store.read_params()
        if "tree_hash" in params and params["tree_hash"] != self.tree.hash:
            raise AssertionError(f"run '{self.directory_name}' was trained on another tree (hash {params['tree_hash']} != {self.tree.hash})")
        for key, value in params.items():
            if key in ("directory_name", "device"):  # where the run lives now, not where it was started
                continue
            self.__dict__[key] = value.to(self.device) if torch.is_tensor(value) else value
        self._barrier()  # every rank has read `params` before rank 0 rewrites it
        if self._rank == 0 and params.get("directory_name") != self.directory_name:
            self._store.write_params(dict(params, directory_name=self.directory_name))
        self.m, self.n = m, n
        saved = self._store.read(m, n, map_location=self.device)
        self.total_steps = saved["total_steps"]
        self.net_params = saved["net_params"]
        for name in ("net", "net_target", "net_reg", "net_reg_"):
            module = self.__new_net()
            module.load_state_dict(saved[name])
            setattr(self, name, module)
        self.optimizer = self.__new_optimizer()
        self.optimizer.load_state_dict(saved["optimizer"])
        self._adopt_optimizer_state()
        # (the checkpoint keeps the reference's keys, so the watch's verdict is not in it: _leaf_watch looks at the first batch it can read
        # instead of waiting for the next multiple of LEAF_CHECK_EVERY -- a resumed run with a sharp policy has its learner within a step)
        self._watch_due = True
        logging.info("resumed at m=%d n=%d (step %d)", m, n, self.total_steps)

    def _adopt_optimizer_state(self):
        """load_state_dict replaces param_groups with the saved ones -- a reference-written checkpoint (rnad.py:232-237, :318) has
        capturable=False / fused=None and keeps `step` as a CPU tensor -- which would leave the captured step and the one-launch
        optimiser tail unusable for the rest of the run.  On a GPU: back to what __new_optimizer constructs, counters on the device."""
        dev = self.device if isinstance(self.device, torch.device) else torch.device(self.device)
        if dev.type != "cuda":
            return
        for group in self.optimizer.param_groups:
            group["capturable"] = True
            group["fused"] = True
            group["foreach"] = False
        for st in self.optimizer.state.values():
            step = st.get("step")
            if step is not None:
                st["step"] = torch.as_tensor(float(step), dtype=torch.float32).to(dev)

    def __save_checkpoint(self):
        if self._rank != 0:
            return
        payload = {name: module.state_dict() for name, module in self._state_nets().items()}
        payload.update(total_steps=self.total_steps, net_params=self.net_params, optimizer=self.optimizer.state_dict())
        self._store.write(self.m, self.n, payload)

    # ------------------------------------------------------------------ schedule (reference learn/rnad.py:321-332)
    def _steps_of_current_update(self):
        """delta_m of regularisation update self.m: bounds[i] is the first m that belongs to phase i + 1.  None: training is over."""
        for bound, steps in zip(self.bounds, self.delta_m):
            if self.m < bound:
                return steps
        return None

    # ------------------------------------------------------------------ evaluation (reference learn/rnad.py:334-351)
    def _evaluate_nashconv(self) -> float:
        """NashConv of the target net over the whole tree (GPU level sweeps, util/metric.py), logged per depth like the reference."""
        data = metric.NashConvData(self.tree)
        data.get_nashconv_from_net(self.tree, self.net_target)
        logging.info("NashConv at m=%d n=%d step %d", self.m, self.n, self.total_steps)
        for depth, mean in data.mean_nashconv_by_depth().items():
            logging.info("  depth %s: %s", depth, mean)
        return (data.row_best[1] + data.col_best[1]).item()

    def _grad_bucket(self, weights):
        """One flat fp32 buffer holding the learner's gradients back to back (the RCCL bucket) and per-tensor views of it."""
        n = sum(w.numel() for w in weights)
        flat = torch.empty((n,), dtype=torch.float32, device=weights[0].device)  # fresh each step: .grad of the last step may
        views, off = [], 0                                                        # still be referenced by the caller
        for w in weights:
            views.append(flat[off: off + w.numel()].view_as(w))
            off += w.numel()
        return flat, views

    def _reg_nets_identical(self):
        """True while net_reg and net_reg_ hold the same weights (all of m == 0).  Checked once per outer iteration."""
        key = (getattr(self, "m", None), id(self.net_reg), id(self.net_reg_),
               sum(p._version for p in self.net_reg.parameters()), sum(p._version for p in self.net_reg_.parameters()))
        if getattr(self, "_reg_identical_key", None) != key:
            a, b = self.net_reg.state_dict(), self.net_reg_.state_dict()
            self._reg_identical = a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
            self._reg_identical_key = key
        return self._reg_identical

    # ------------------------------------------------------------------ tabular evaluation of the four nets
    def _fused_mlp(self):
        A = self.tree.max_actions
        return isinstance(self.net, net.MLP) and self.net._fusable() and rnad_hip.mlp_backward_supported(A, self.net.width)

    def _tabular_mode(self, T, B):
        """RNaD.tabular if the tree is small enough next to a [T, B] trajectory for the table evaluation to pay, else False."""
        mode = getattr(self, "tabular", False)
        if not mode or not self._fused_mlp():
            return False
        if getattr(self, "tabular_gate", 8) * self.tree.handle().S > T * B:
            return False
        # the per-row sums are fixed point with headroom for one addend per lane: 2^22 lanes per call in the bucketed learner
        # (csrc/bucket.hip kLaneBits), 2^21 in the round-1 global-atomics kernel that non-bucketable trees fall back to
        if mode is True and (B > rnad_hip.BUCKET_MAX_LANES or (B > 2**21 and rnad_hip.bucket_plan(self.tree.handle(), B) is None)):
            return "forward"
        return mode

    def _known_norm(self, T):
        """N_P of the GLOBAL batch without a collective (f64 [2] on the device), or None.  On a tree whose episodes all last 2 * max_depth
        env steps (TreeHandle.uniform_length) every lane is alive at every step of the window: N_0 = N_1 = batch_size * T / 2, whatever the
        ranks played -- the all-reduce of the normalisers (vtrace.py:373,388 are sums over the whole batch) has nothing to add.
        RNaD.analytic_norm = False keeps the collective."""
        handle = self.tree.handle()
        if not (self._dp() and getattr(self, "analytic_norm", True) and handle.uniform_length and T == 2 * handle.max_depth):
            return None
        key = (self.batch_size, T)
        cached = self.__dict__.get("_known_norm_cache")
        if cached is None or cached[0] != key:
            n = float(self.batch_size // self._world * self._world) * T / 2
            cached = self._known_norm_cache = (key, torch.tensor([n, n], dtype=torch.float64, device=self.device))
        return cached[1]

    def _fuse_now(self):
        """RNaD.fuse_rollout_learner (default on; RNAD_FUSE_PLAY_LEARN=0 turns the default off): rollout and learner of the step in one launch."""
        # (data parallel: off unless asked for -- with two launches the all-reduce of the normalisers runs beside the learner; behind the
        # one launch it would sit, exposed, between it and the finish: a collective's latency for the 6 us the fusion saves -- except on
        # trees whose normalisers need no collective, _known_norm)
        default = os.environ.get("RNAD_FUSE_PLAY_LEARN", "1") != "0" and (
            not self._dp() or self._known_norm(2 * self.tree.handle().max_depth) is not None)
        return bool(getattr(self, "fuse_rollout_learner", default))

    DISTINCT_AFTER = 4096  # updates after which `distinct_trajectories = None` switches the learner to the distinct trajectories of a work item

    def _distinct_now(self):
        """RNaD.distinct_trajectories: the learner half of the one-launch rollout + learner runs once per distinct trajectory of a work item
        (rnad_hip.rollout_learn_bucketed_compact(distinct=True): the same per-row sums bit for bit).  It pays once lanes pile up on few
        trajectories -- a trained policy -- and costs ~4 % under the near-uniform policies of fresh nets: None
        (default) turns it on after DISTINCT_AFTER updates of this trainer; True / False force it."""
        want = getattr(self, "distinct_trajectories", None)
        if want is not None:
            return bool(want)
        # (r05: or earlier, once the watch over the bucket sizes has seen the lanes pile up -- sticky, like its other verdict)
        return self.total_steps >= self.DISTINCT_AFTER or self.__dict__.get("_distinct_crowded", False)

    def _leaf_now(self, handle, local_batch, T_cap):
        """RNaD.leaf_paths (None: automatic; True / False force it; RNAD_LEAF_PATHS=0 / 1 overrides): the learner of the one-call rollout +
        learner runs on the tree's LEAF PATHS (rnad_hip.LeafPaths: one column per terminal transition, weighted with the lanes the rollout
        counted on it) instead of once per lane -- the same per-row sums bit for bit (integer sums), as two launches.  Its cost is the
        tree's, not the batch's.  Automatic when every episode has the tree's full length (so that every lane leaves the tree inside the
        window), the rank plays at least two lanes per leaf path (measured on configs[1]'s 531 441 paths, uniform policies: a tie with the
        one-launch rollout + learner at 2^20 lanes, 250 -> 220 us per step at 2^21, 403 -> 311 at 2^22).  r06: a CROWDED bucket no
        longer ends it.  A work item of the leaf learner used to count all lanes of its bucket, so a bucket that a sharpened policy fills
        with ten times its even share made the step 2 - 2.5 times slower (x40 policy head: 0.269 / 0.439 / 0.769 ms at 2^20 / 2^21 / 2^22
        lanes) and RNaD._leaf_watch switched the leaf learner off for good; now the rollout's work items count such a bucket themselves
        (rnad_leaf_paths_t.col_count: a histogram in LDS, a global atomic per non-zero bin) and the same steps take 0.160 / 0.209 / 0.298 ms,
        within 7 % of the learner on the distinct trajectories of a work item (0.147 / 0.196 / 0.289), with unchanged times under uniform
        policies (profiles/r06_leaf_sharp.md)."""
        want = getattr(self, "leaf_paths", None)
        env = os.environ.get("RNAD_LEAF_PATHS")
        if env is not None:
            want = env != "0"
        if want is False or not handle.uniform_length or T_cap != 2 * handle.max_depth or T_cap > rnad_hip.COMPACT_MAX_STEPS:
            return None
        if want is None:
            n = handle.__dict__.get("_n_terminal")
            if n is None:
                live = (self.tree.index_tensor == 0) & (self.tree.chance_tensor > 0)
                n = handle._n_terminal = int(live[1:].sum().item())
            if 2 * n > local_batch:
                return None
        return rnad_hip.leaf_paths(handle, local_batch, self.tree.index_tensor, self.tree.chance_tensor, self.tree.value_tensor)

    LEAF_CHECK_EVERY = 1024  # steps between two looks at the bucket sizes (a sync and a small device -> host copy: ~0.3 ms)
    LEAF_CROWDED = 4.0       # (r05: ended the automatic leaf learner; r06: rnad_hip.LEAF_CROWDED_SHARE -- from this share on the rollout counts a bucket's lanes)
    DISTINCT_CROWDED = 4.0   # a bucket with more than this many times its even share turns the automatic learner on distinct trajectories on before DISTINCT_AFTER
    # (profiles/r05_leaf.md, 2^21 lanes: at a share of 3.4 the leaf learner is still ahead of the per-lane one, 0.233 against 0.262 ms per
    # step, at 6.1 behind it, 0.275 against 0.266; the distinct trajectories of a work item take 0.259 / 0.239 / 0.223 / 0.214 at shares
    # of 2.06 (fresh nets) / 3.1 / 3.4 / 6.1.  With the reference's lr = 5e-5 the share drifts between 2 and 3.5 over the first 6 000
    # updates -- tools/micro/share_probe.py --, so neither switch falls into a benchmark of fresh nets)

    def _leaf_watch(self):
        """The automatic learner's look at the batch (see _distinct_now): every LEAF_CHECK_EVERY steps the work list of the last batch is
        read back -- lanes per bucket -- and compared with the even share: above DISTINCT_CROWDED the learner on the distinct trajectories
        of a work item comes on before DISTINCT_AFTER (sticky: policies sharpen).  A check that falls on a step whose batch cannot be read
        (a logging step, another rollout) is retried on the next step that can.  (r05 also ended the leaf-path learner here, for good; r06:
        the rollout counts crowded buckets itself, _leaf_now, and the leaf learner stays.)"""
        if (getattr(self, "distinct_trajectories", None) is not None or self.__dict__.get("_distinct_crowded", False)
                or self.total_steps >= self.DISTINCT_AFTER):
            return
        if self.total_steps % self.LEAF_CHECK_EVERY == self.LEAF_CHECK_EVERY - 1:
            self._watch_due = True
        if not self.__dict__.get("_watch_due", False):
            return
        ep = self.__dict__.get("last_episodes")
        buckets = getattr(ep, "buckets", None) if ep is not None else None
        if buckets is None or getattr(ep, "_compact", None) is None:
            return  # (not the compact bucketed step: retried on the next one)
        self._watch_due = False
        if not self.tree.handle().uniform_length:
            # (a ragged tree, whose buckets differ in size by construction: "times the even share" says nothing there)
            return
        n = int(buckets.n_items.item())
        items = buckets.items[:n].cpu()  # (a few thousand work items: counted on the host -- no kernel of torch's that the step has not loaded yet)
        per_bucket = torch.bincount(items[:, 2].long(), weights=items[:, 1].double(), minlength=buckets.plan.n_buckets)
        share = float(per_bucket.max().item()) * max(buckets.plan.n_groups, 1) / max(ep.batch_size, 1)
        if self._dp():  # every rank decides alike (they capture, or drop, the same graph)
            t = torch.tensor([share], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            share = float(t.item())
        self._leaf_share = share
        if share > self.DISTINCT_CROWDED:
            logging.info("learner on distinct trajectories on: a bucket holds %.1f times its even share of the lanes", share)
            self._distinct_crowded = True

    def _learn_params(self, alpha):
        return rnad_hip.make_learn_params(
            alpha=alpha, eta=self.eta, lambda_=1.0, c=self.c_bar, rho=self.roh_bar, gamma=self.vtrace_gamma,
            clip=self.neurd_clip, threshold=self.beta, w_v=self.value_weight, w_n=self.neurd_weight,
            eps_threshold=self.epsilon_threshold, n_disc=self.n_discrete)

    def _reg_tables(self, table, fold=False):
        """Logits of net_reg and net_reg_ on the tree's 2S observations.  The regularisation nets are constant between two
        rotations (rnad.py:528-531), so their tables are evaluated when their weights change (tensor version counters: bumped by
        load_state_dict and by in-place edits under no_grad) and kept, in place: the buffers keep their addresses for as long as
        the observation table does (a captured graph of the step reads them).  Call invalidate_tables() after editing the nets
        in a way autograd's version counters do not see."""
        key = (id(self.net_reg), id(self.net_reg_), table.data_ptr(), tuple(table.shape),
               sum(p._version for p in self.net_reg.parameters()), sum(p._version for p in self.net_reg_.parameters()), bool(fold))
        cache = getattr(self, "_reg_table_cache", None)
        if cache is None or cache["ident"] != key[:4]:
            cache = self._reg_table_cache = {"key": None, "ident": key[:4], "logit_reg": None, "logit_reg_": None}
        if cache["key"] != key:
            A = self.tree.max_actions
            with torch.no_grad():
                outs = rnad_hip.mlp_forward_multi(rnad_hip.mlp_pack_many([self.net_reg._weights(), self.net_reg_._weights()], A, fold=fold),
                                                  self.net.width, table, A,
                                                  [(True, False), (True, False)], fold=self.tree.handle() if fold else False)
            for name, out in (("logit_reg", outs[0][0]), ("logit_reg_", outs[1][0])):
                if cache[name] is None:
                    cache[name] = out
                else:
                    cache[name].copy_(out)
            cache["key"] = key
        return cache["logit_reg"], cache["logit_reg_"]

    def invalidate_tables(self):
        """The hook for weight edits torch's version counters do not see (`.data` writes, dist.broadcast(t.data), raw-pointer kernels), on
        ANY of the four nets: the next step re-evaluates the regularisation tables (net_reg, net_reg_) and re-packs the weight images of
        net and net_target -- all IN PLACE: the buffers keep their addresses, which a captured graph of the step has baked in (a fresh
        allocation would leave the replays reading the old ones)."""
        cache = getattr(self, "_reg_table_cache", None)
        if cache is not None:
            cache["key"] = None
        for entry in self.__dict__.get("_packed_cache", {}).get("layouts", {}).values():
            entry["key"] = None

    def _fold(self):
        """The table evaluations of the per-row mode use the FOLD kernels: asked for, and the tree's observation table allows it."""
        return bool(getattr(self, "fold_legal", True)) and self.tree.handle().legal_foldable

    def _packed_images(self, fold=None):
        """(image of net, image of net_target): the packed weight layouts the fused MLP kernels read (rnad_hip.mlp_pack), in persistent
        buffers -- one pair per layout (plain / FOLD).  The pair the one-launch optimiser tail maintains (rnad_optimizer_step writes
        every new weight into the tensor AND its image slot: a training step carries no pack launch) is re-packed only when the nets'
        tensors changed as far as torch can tell (version counters, data pointers); the other pair on every request.
        fold: which layout (default: the maintained one)."""
        A = self.tree.max_actions
        ws = (self.net._weights(), self.net_target._weights())
        cache = self.__dict__.setdefault("_packed_cache", {"layouts": {}, "maintained": None})
        if fold is None:
            fold = bool(cache["maintained"])
        fold = bool(fold)
        key = tuple((id(w), w.data_ptr(), w._version) for group in ws for w in group)
        shape = (A, self.net.width, ws[0][0].device)
        entry = cache["layouts"].get(fold)
        if entry is None or entry["shape"] != shape:
            size = rnad_hip.mlp_packed_size(A, self.net.width, fold)
            entry = cache["layouts"][fold] = {"shape": shape, "key": None,
                                              "images": [torch.empty((size,), dtype=torch.float32, device=ws[0][0].device) for _ in range(2)]}
        if entry["key"] != key or cache["maintained"] != fold:
            rnad_hip.mlp_pack_many(list(ws), A, out=entry["images"], fold=fold)
            entry["key"] = key
        return entry["images"]

    def _table_outputs(self, alpha, obs_half=False, want_target_logits=False, policy_only=False, fold=False, records_hp=None,
                       step_params=None, shard=False, dedup=None, defer_expand=False):
        """learner / target / regularisation nets on the 2S observations of the tree (rnad.py:373-380 on every distinct input):
        learner and target in ONE launch per step, the two regularisation nets from _reg_tables.  Both regularisation tables are
        always there: a term of log_policy_reg (:382) whose weight is exactly 0 adds exactly 0.
        policy_only: the learner's logits alone (what the rollout needs); _value_tables adds the value heads on the visited rows."""
        A = self.tree.max_actions
        table = self.tree.handle().observations_table(obs_half)
        packed, packed_target = self._packed_images(fold)
        self._layout_in_use = bool(fold)  # (what the optimiser tail of this step will keep current)
        if policy_only:
            # lazy rows: the learner's policy head is evaluated in stages by the rollout (staged_actor below: the upper rows of the cut,
            # then the rows of the groups the batch descends into); rows no lane can reach stay uninitialised and are never read
            logit = torch.empty((table.shape[0], A), dtype=torch.float32, device=table.device)
            # the actor's policy rows come out of the same launch as its logits (rnad_mlp_forward_actor): the rollout kernels gather from them
            logit._policy_rows = torch.empty((table.shape[0], int(rnad_hip.lib().rnad_bucket_policy_row_stride(A))), dtype=torch.float32,
                                             device=table.device)

            def staged_actor(rows, packed=packed, logit=logit, table=table, width=self.net.width, fold=self.tree.handle() if fold else False):
                with torch.no_grad():
                    rnad_hip.mlp_forward_actor(self.tree.handle(), packed, width, table, logit, logit._policy_rows, rows=rows, fold=fold)

            logit_reg, logit_reg_ = self._reg_tables(table, fold)
            return dict(table=table, logit=logit, v=None, logit_target=None, v_target=None, logit_reg=logit_reg, logit_reg_=logit_reg_,
                        packed_net=packed, packed_target=packed_target, staged_actor=staged_actor, fold=fold)
        if records_hp is not None and not want_target_logits and rnad_hip.mlp_rows_records_supported(A, self.net.width, fold):
            # records_hp: the caller wants the row records of this step too (rnad_hip.bucket_records(fast=True)) -- forwards and records
            # come out of ONE launch (csrc/mlp_rows.hip: a persistent workgroup per CU, a wave per hidden tile, weights in registers)
            logit_reg, logit_reg_ = self._reg_tables(table, fold)
            rows, per = self._row_shard(self.tree.handle()) if shard else (None, 0)
            if dedup is not None:
                rows = dedup.uniq  # one representative row per distinct observation
            with torch.no_grad():
                out = rnad_hip.mlp_rows_records(self.tree.handle(), packed, packed_target, self.net.width, table, logit_reg, logit_reg_,
                                                records_hp, step_params=step_params, fold=self.tree.handle() if fold else False, rows=rows,
                                                alloc_rows=per * self._world if shard else (table.shape[0] if dedup is not None else None))
                if dedup is not None:
                    # every other row: a copy of its representative's records (the same bits the launch on all rows writes) -- made by the
                    # keys pass of the rollout that follows (rnad_rollout_bucketed_compact_expand), or right here when none does
                    # r05: of the three tables only the fast records and the policy rows travel with the step (11.6 of the 19 MB: the
                    # rollout and the learner gather those); the 64-byte row records of the other rows are copied when somebody reads
                    # them (rnad_hip.complete_records: the dense views of the batch, a logging step)
                    if defer_expand and out["policy_rows"] is not None:
                        out["records"]._expand = (dedup, [out["fast_records"], out["policy_rows"]])
                        out["records"]._expand_job, out["records"]._expand_stale = (dedup, [out["records"]]), True
                    else:
                        rnad_hip.rows_expand(dedup, [out["fast_records"], out["policy_rows"], out["records"]])
            tables = dict(table=table, logit=out["logit"], v=out["v"], logit_target=None, v_target=out["v_target"], logit_reg=logit_reg,
                          logit_reg_=logit_reg_, packed_net=packed, fold=fold, records=out["records"], fast_records=out["fast_records"],
                          dedup=dedup)
            if shard:
                # this rank evaluated its rows only: the actor's policy rows first (the rollout needs them), then the learner's operands
                self._gather_rows(out["policy_rows"], per)
                self._gather_rows(out["fast_records"], per)
                self._gather_rows(out["records"], per)  # (only read by logging / Episodes' dense views: DESIGN.md section 7)
                N = table.shape[0]  # (the shares are padded to equal size: the tables proper are the first 2S rows)
                tables["records"], tables["fast_records"] = out["records"][:N], out["fast_records"][:N]
                tables["records"]._policy_rows = out["policy_rows"][:N]
                for name in ("logit", "v", "v_target"):
                    tables[name] = tables[name][:N]
                tables["shard_rows"] = rows
            return tables
        with torch.no_grad():
            # (one launch entry per (net, head) -- three equal work units per 64-row span -- was measured: 45.5 instead of 43.4 us, every
            # workgroup loads its net's 43 KB weight image first)
            outs = rnad_hip.mlp_forward_multi([packed, packed_target], self.net.width, table, A,
                                              [(True, True), (want_target_logits, True)], fold=self.tree.handle() if fold else False)
        logit_reg, logit_reg_ = self._reg_tables(table, fold)
        tables = dict(table=table, logit=outs[0][0], v=outs[0][1], logit_target=outs[1][0], v_target=outs[1][1], logit_reg=logit_reg,
                      logit_reg_=logit_reg_, packed_net=packed, fold=fold)
        if records_hp is not None:
            tables["records"], tables["fast_records"] = rnad_hip.bucket_records(
                self.tree.handle(), tables["logit"], tables["v"], tables["v_target"], logit_reg, logit_reg_, records_hp,
                step_params=step_params, fast=True)
        return tables

    def _value_tables(self, tables, visited, alpha, step_params=None):
        """Lazy rows, after the rollout: the learner's and the target's value heads, the row records and (in __learn) the gradient
        tables and the backward on the rows the batch visited -- `visited` int32 [2S] from the rollout, compacted on the stream."""
        handle, A = self.tree.handle(), self.tree.max_actions
        rows = rnad_hip.compact_valid(visited)
        if rnad_hip.mlp_rows_records_supported(A, self.net.width, tables.get("fold", False), True):
            # both value heads on the listed rows and their records in one launch (csrc/mlp_rows.hip, the variant that reads the logits)
            with torch.no_grad():
                out = rnad_hip.mlp_rows_records(handle, tables["packed_net"], tables["packed_target"], self.net.width, tables["table"],
                                                tables["logit_reg"], tables["logit_reg_"], self._learn_params(alpha), step_params=step_params,
                                                fold=handle if tables.get("fold", False) else False, rows=rows, logit_tab=tables["logit"])
            tables["v"], tables["v_target"] = out["v"], out["v_target"]
            tables["records"], tables["fast_records"] = out["records"], out["fast_records"]
            tables["rows"] = rows
            self.last_rows = rows
            return tables
        with torch.no_grad():
            # (the rows that are not listed are never read: records, gradient tables and the backward all go by the same list)
            fold = handle if tables.get("fold", False) else False
            if fold:  # both value heads in one launch
                outs = rnad_hip.mlp_forward_multi([tables["packed_net"], tables["packed_target"]], self.net.width, tables["table"], A,
                                                  [(False, True), (False, True)], fold=fold, live=rows, zero_rest=False)
                tables["v"], tables["v_target"] = outs[0][1], outs[1][1]
            else:
                tables["v"] = rnad_hip.mlp_forward(tables["packed_net"], self.net.width, tables["table"], A, want_logits=False, live=rows,
                                                   zero_rest=False)[1]
                tables["v_target"] = rnad_hip.mlp_forward(tables["packed_target"], self.net.width, tables["table"], A, want_logits=False, live=rows,
                                                          zero_rest=False)[1]
        tables["records"], tables["fast_records"] = rnad_hip.bucket_records(
            handle, tables["logit"], tables["v"], tables["v_target"], tables["logit_reg"], tables["logit_reg_"], self._learn_params(alpha),
            step_params=step_params, fast=True, rows=rows)
        tables["rows"] = rows
        self.last_rows = rows  # (bench.py reads how many rows a step visited)
        return tables

    def _use_lazy_rows(self, handle, local_batch, T_cap, log, buffer):
        want = getattr(self, "lazy_rows", None)
        if want is None:
            want = 2 * handle.S > local_batch
        return bool(want and log is None and getattr(self, "compact_trajectory", True) and T_cap <= rnad_hip.COMPACT_MAX_STEPS
                    and self.buffer_mod == 1 and buffer.max_size == 1 and not getattr(self, "store_actor_values", False)
                    and rnad_hip.bucket_plan(handle, local_batch) is not None)

    # ------------------------------------------------------------------ reference learn/rnad.py:353-456
    @staticmethod
    def _logits_of(module, episodes, want_logits=True, want_value=True, live=None, table=None):
        """Raw policy logits [T*B, A] and value [T*B, 1] of `module` on a trajectory (a head that is not wanted may be None).
        live: evaluate only the (t, b) slots of that rnad_hip.LiveRows list (zeros elsewhere).
        table: evaluate on these [2S, 2, A, A] observations (one per player and state) instead of the trajectory's."""
        T = episodes.t_eff + 1
        if table is not None:
            return module.forward_logits(table, want_logits=want_logits, want_value=want_value)
        if hasattr(module, "forward_logits"):
            kw = {"live": live} if live is not None else {}
            return module.forward_logits(episodes.observations[:T], want_logits=want_logits, want_value=want_value, **kw)
        logit, _, _, v = module.forward_batch(episodes)  # any module honouring the reference contract (nn/net.py:64-85)
        A = logit.shape[-1]
        return logit.reshape(-1, A), v.reshape(-1, 1)

    def __learn(self, episodes: episode.Episodes, alpha: float, log: dict = None, tables: dict = None, defer_clip: bool = False):
        """Gradients of the learner net from a batch of trajectories (reward transform + V-trace + NeuRD).
        tables: _table_outputs(alpha) of the CURRENT nets, when train_step already evaluated them for the rollout.
        defer_clip: leave clip_grad_norm_ (rnad.py:456) to the fused optimiser kernel of train_step; the flat gradient bucket is
        then handed over in self._pending_flat."""
        T, B, A = episodes.t_eff + 1, episodes.batch_size, self.tree.max_actions

        # N_P = #(valid & turn == P): batch-global loss normalisers (vtrace.py:373,388).  Their all-reduce over the ranks is
        # issued first and overlaps the MLP forwards below (RCCL runs it on its own stream).
        norm = episodes.norm_for_learner() if hasattr(episodes, "norm_for_learner") else episodes.valid_counts
        norm_work = None
        known = (self._known_norm(T) if (self._dp() and getattr(episodes, "buckets", None) is not None
                                        and B == getattr(self, "batch_size", -1) // self._world) else None)
        if known is not None:
            norm = known  # (a uniform-length tree: the global counts are a constant)
        elif self._dp():
            norm = norm.clone()  # the all-reduce is in place, and the episodes keep their own count
            norm_work = dist.all_reduce(norm, async_op=True)

        reuse = (getattr(self, "reuse_actor_outputs", False) and getattr(episodes, "actor_logits", None) is not None
                 and getattr(episodes, "_actor_tag", None) == (id(self.net), self.total_steps)
                 and rnad_hip.mlp_backward_supported(A, getattr(self.net, "width", 0)))
        # Ragged trajectories (pruned trees, padded replay batches): the reference evaluates all four nets on every (t, b) slot
        # and masks the absorbed ones afterwards (valid, :369).  Here the nets run on the live slots only; the others hold
        # zeros, which the same masks discard -- losses and gradients are unchanged.  Logging steps of the dense mode evaluate
        # every slot, because logit_mean / logit_max (:427-452) are taken over ALL slots.
        fused_mlp = self._fused_mlp()
        # Tabular evaluation (RNaD.tabular): an observation depends on (state, player to move) only, so each net is evaluated on
        # the 2S distinct observations of the tree and every (t, b) slot gathers its row (include/rnad_hip.h,
        # rnad_learn_fused_gather / rnad_learn_fused_tabular / rnad_learn_bucketed).  Worth it when the tree is small next to the
        # batch (configs[1]: 132 862 rows for 12.6 M slots).
        mode = self._tabular_mode(T, B)
        table = None
        if mode:
            if tables is None:
                tables = self._table_outputs(alpha, getattr(episodes, "obs_half", False), want_target_logits=log is not None,
                                             fold=mode is True and self._fold())
            table = tables["table"]
        per_row_backward = table is not None and mode is True
        bucketed = per_row_backward and getattr(episodes, "buckets", None) is not None
        if per_row_backward and not bucketed and B > 2**21:
            per_row_backward = False  # a batch that is not bucket-ordered (a replay sample) beyond the atomics kernel's 2^21 lanes: per-slot backward
        if not bucketed and getattr(episodes, "buckets", None) is not None:
            rnad_hip.bucket_alive(self.tree.handle(), episodes.buckets)  # (a no-op unless the rollout left its alive counts to the compact learner)
        if table is not None and not bucketed and (tables.get("dedup") is not None or tables.get("shard_rows") is not None):
            # (_plays_what_it_learns keeps _step_body from getting here; a caller who hands such tables in with another batch is told)
            raise RuntimeError("these tables hold the nets' outputs for the representatives of the distinct observations (or for one rank's "
                               "rows) only: a batch that is not bucket-ordered gathers logit / v / v_target from EVERY row -- evaluate "
                               "the tables on all rows for it (_table_outputs without dedup / shard)")
        live, capacity = None, None
        if (not per_row_backward and getattr(self, "skip_absorbed", True) and log is None and fused_mlp
                and not self.tree.handle().uniform_length):
            live = rnad_hip.compact_valid(episodes.indices[:T])
        fwd_live = None if table is not None else live
        # The fused MLP is differentiated by hand (rnad_mlp_backward), so the learner's forward needs no autograd graph and its
        # gradients can be written straight into one flat bucket (the all-reduce buffer).  Any other net goes through autograd.
        direct = fused_mlp
        reuse = reuse and table is None
        if table is not None:
            logit, v, logit_target, v_target = tables["logit"], tables["v"], tables["logit_target"], tables["v_target"]
            logit_reg, logit_reg_ = tables["logit_reg"], tables["logit_reg_"]
            if log is not None and logit_target is None:
                logit_target = self.net_target.forward_logits(table, want_value=False)[0]
        elif reuse:  # the rollout's own outputs: same weights, same observations, same kernel -> same bits as rnad.py:373
            logit, v = episodes.actor_logits.reshape(-1, A), episodes.values[:T].reshape(-1, 1)
        elif direct:
            with torch.no_grad():
                logit, v = self._logits_of(self.net, episodes, live=fwd_live)  # rnad.py:373
        else:
            logit, v = self._logits_of(self.net, episodes, live=fwd_live)  # rnad.py:373, with grad
        if table is None:
            with torch.no_grad():
                # the reference runs all four full nets (:378-380); only these heads are ever read (:382-406)
                logit_target, v_target = self._logits_of(self.net_target, episodes, want_logits=log is not None, live=fwd_live)  # :378
                # log_policy_reg = log_pi - (alpha * log_pi_reg + (1 - alpha) * log_pi_reg_) (:382).  A term whose weight is exactly
                # 0 adds exactly 0 (log-policies are finite), and two nets with the same weights give the same bits: in the second
                # half of every outer iteration (alpha == 1, :497) and during all of m == 0 (both reg nets are copies of the initial
                # net, :183-186) one evaluation serves both operands.
                if alpha == 0:
                    logit_reg_, _ = self._logits_of(self.net_reg_, episodes, want_value=False, live=fwd_live)  # :380
                    logit_reg = logit_reg_
                else:
                    logit_reg, _ = self._logits_of(self.net_reg, episodes, want_value=False, live=fwd_live)  # :379
                    if alpha == 1 or self._reg_nets_identical():
                        logit_reg_ = logit_reg
                    else:
                        logit_reg_, _ = self._logits_of(self.net_reg_, episodes, want_value=False, live=fwd_live)  # :380

        # the bucketed learner needs the normalisers only in its last kernel: their all-reduce runs beside k_bucket_learn
        # (row sharding finishes late as well: the ranks' per-row sums are all-reduced between the learner and the finish)
        late_norm = bucketed and (norm_work is not None or (tables is not None and tables.get("shard_rows") is not None))
        if norm_work is not None and not late_norm:
            norm_work.wait()
        hp = self._learn_params(alpha)
        if bucketed:
            # bucket-ordered batch (Episodes.generate(bucketed=True)): per-row sums in LDS, no global atomics (csrc/bucket.hip)
            records = tables.get("records")
            if records is None:
                records = rnad_hip.bucket_records(self.tree.handle(), logit, v, v_target, logit_reg, logit_reg_, hp)
            dedup, grouped, rows_now = tables.get("dedup"), None, tables.get("rows")
            compact = getattr(episodes, "_compact", None)
            if compact is not None and compact[1] is records and tables.get("fast_records") is not None:
                # the batch was played this very step with the pi columns of these records as the actor: 64 bytes per lane
                if (dedup is not None and getattr(self, "group_sums_in_finish", True)
                        and dedup.groups_below_cut(self.tree.handle(), episodes.buckets.plan)):
                    # distinct observations: k_bucket_finish adds the rows of a group up into its representative's row -- no pass over
                    # the gradient tables between finish and the backward
                    grouped, rows_now = dedup, dedup.singles
                learned = episodes.__dict__.pop("_learned", None)
                if learned is not None and learned["records"] is records and log is None and T == compact[0].T_cap:
                    # the rollout's own launch added this batch's update up (Episodes.generate(learn=...), k_bucket_play_learn) and,
                    # single process, finished it; data parallel: the finish below, once the normalisers are all-reduced
                    dlogit, dv, losses = learned["dlogit"], learned["dv"], None
                    if dlogit is None:
                        S2 = 2 * self.tree.handle().S
                        dlogit = torch.empty((S2, A), dtype=torch.float32, device=records.device)
                        dv = torch.empty((S2, 1), dtype=torch.float32, device=records.device)
                        late_norm = True  # (the finish below; nothing to wait for when the normalisers are known)
                else:
                    assert learned is None, "a batch whose update rode in its rollout can only be learned from as it was played"
                    dlogit, dv, losses = rnad_hip.learn_bucketed_compact(self.tree.handle(), episodes.buckets, compact[0], T, records,
                                                                         tables["fast_records"], None if late_norm else norm, hp,
                                                                         want_losses=log is not None, rows=rows_now, groups=grouped)
                live = rows_now  # lazy rows: the backward runs on the visited rows only
            else:
                rnad_hip.bucket_alive(self.tree.handle(), episodes.buckets)  # (a no-op unless the rollout left its counts to a compact learner)
                rnad_hip.complete_records(records)
                dlogit, dv, losses = rnad_hip.learn_bucketed(self.tree.handle(), episodes.buckets, episodes.indices[:T], episodes.action_idx[:T],
                                                             episodes.rewards[:T], episodes.policy[:T], records,
                                                             None if late_norm else norm, hp, want_losses=log is not None)
            shard = tables.get("shard_rows") if late_norm else None
            if shard is not None:
                # row sharding: the 64-bit per-row sums of every rank's lanes -> the sums of the global batch on every rank (integer
                # addition: exact, the same bits whatever the order); this rank finishes and back-propagates its own rows
                plan, A1 = episodes.buckets.plan, A + 1
                S_ = self.tree.handle().S
                dist.all_reduce(plan.accumulators[: 2 * S_ * A1 + rnad_hip.BUCKET_REPLICAS * 2 * max(plan.n_upper, 1) * A1])
            if late_norm:
                if norm_work is not None:
                    norm_work.wait()
                rnad_hip.bucket_finish(self.tree.handle(), episodes.buckets, norm, hp, dlogit, dv, losses,
                                       rows=shard if shard is not None else rows_now, groups=grouped)
            if shard is not None:
                plan.accumulators[: 2 * S_ * A1].zero_()  # (finish cleared the rows it read; the other ranks' rows still hold their sums)
                live = shard
            if getattr(self, "keep_last_tables", False):  # (tests: the per-row gradient tables of this step)
                self.last_tables = (dlogit.clone(), dv.clone(), shard)
            if dedup is not None and shard is None:
                # rows with the same observation: their dL/dlogit, dL/dv added up into the representative row; the backward on those
                if grouped is None:
                    rnad_hip.rows_segment_sum(dedup, A, dlogit, dv)
                live, capacity = dedup.uniq, dedup.n_unique
            pi = None
            backward_obs = table
        elif table is not None:
            fn = rnad_hip.learn_fused_tabular if per_row_backward else rnad_hip.learn_fused_gather
            dlogit, dv, losses = fn(self.tree.handle(), episodes.indices[:T], episodes.mask_bits[:T], episodes.action_idx[:T],
                                    episodes.rewards[:T], episodes.policy[:T], logit, v, v_target, logit_reg, logit_reg_, norm, hp)
            pi = None
            # per-row: dlogit [2S, A], dv [2S, 1] are sums of the per-slot gradients; else they are per slot, as in the dense path
            backward_obs = table if per_row_backward else episodes.observations[:T]
        else:
            dlogit, dv, losses, pi, _, _ = rnad_hip.learn_fused(
                episodes.indices[:T], episodes.mask_bits[:T], episodes.action_idx[:T], episodes.rewards[:T], episodes.policy[:T],
                logit.detach().contiguous(), v.detach().reshape(T, B).contiguous(), v_target.reshape(T, B).contiguous(),
                logit_reg.contiguous(), logit_reg_.contiguous(), norm, hp, want_aux=log is not None)
            backward_obs = episodes.observations[:T]
        # loss.backward() (rnad.py:424-425) with the closed-form dL/dlogit, dL/dv
        flat = None
        if reuse or direct:
            weights = self.net._weights()
            flat, views = self._grad_bucket(weights)
            fold = tables is not None and tables.get("fold", False)
            if fold and backward_obs is not table:  # (per-slot backward of a batch that could not take the per-row path: unfolded image)
                packed, fold = self.net.pack(), False
            else:
                packed = tables["packed_net"] if tables is not None and "packed_net" in tables else self.net.pack()  # same weights as the forward
            rnad_hip.mlp_backward(packed, weights, backward_obs, A, dlogit.view(-1, A), dv.view(-1, 1), live=live, out=views,
                                  fold=self.tree.handle() if fold else False, capacity=capacity)
            if all(p_.grad is None for p_ in weights):
                for p_, g_ in zip(weights, views):
                    p_.grad = g_
            else:  # gradient accumulation across calls, as loss.backward() would
                for p_, g_ in zip(weights, views):
                    p_.grad = g_.clone() if p_.grad is None else p_.grad + g_
                flat = None
        else:
            torch.autograd.backward([logit, v], [dlogit.view(-1, A), dv.view(-1, 1)])

        if self._dp():
            if flat is not None:
                dist.all_reduce(flat)  # one 43 KB bucket over RCCL, in place: the .grad tensors are views of it
            else:
                grads = [p.grad for p in self.net.parameters()]
                cat = torch.cat([g.reshape(-1) for g in grads])
                dist.all_reduce(cat)
                off = 0
                for g in grads:
                    g.copy_(cat[off: off + g.numel()].view_as(g))
                    off += g.numel()
            if log is not None:
                dist.all_reduce(losses)

        if log is not None:
            if table is not None:
                # the logged statistics are over per-slot tensors (rnad.py:427-452): gather them from the tables.  An absorbed
                # slot gathers the row of state 0 -- the reference's net output there, as it evaluates the net on state 0's
                # observation -- so the statistics are those of the dense path.
                S_ = self.tree.handle().S
                rows = (episodes.indices[:T].long()
                        + (torch.arange(T, device=logit.device) & 1).view(T, 1) * S_).reshape(-1)
                logit = logit.index_select(0, rows)
                logit_target = logit_target.index_select(0, rows)
                pi = rnad_hip.policy_head(logit.contiguous(), mask_bits=episodes.mask_bits[:T].reshape(-1)).view(T, B, A)
            total_norm = 0
            for p in self.net.parameters():
                total_norm += p.grad.detach().data.norm(2).item() ** 2
            total_norm = total_norm**0.5
            valid = (episodes.indices[:T] != 0).to(torch.float)
            masks = episodes.masks[:T]
            logit_mean = logit.mean().item()
            pi_target = rnad_hip.policy_head(logit_target.contiguous(), mask_bits=episodes.mask_bits[:T].reshape(-1)).view(T, B, A)
            uniform_policy = torch.nn.functional.normalize(masks, p=1, dim=-1)
            log.update({
                "loss_v": losses[0].item(),
                "loss_nerd": losses[1].item(),
                "traj_len": valid.sum(0).mean(-1).item(),
                "gradient_norm": total_norm,
                "logit_mean": logit_mean,
                "logit_max": torch.max(torch.abs(logit - logit_mean)).item(),
                "entropy": metric.kld(pi, uniform_policy, valid, legal_actions=masks),
                "entropy_target": metric.kld(pi_target, uniform_policy, valid, legal_actions=masks),
                "actor_learner_kld": metric.kld(pi, episodes.policy[:T], valid, legal_actions=masks),
            })

        self._pending_flat = None
        if defer_clip and flat is not None and flat.is_cuda:
            self._pending_flat = flat  # clipped by rnad_optimizer_step together with Adam and the EMA
        elif flat is not None and flat.is_cuda:  # the .grad tensors are views of this bucket: one launch instead of ~8
            rnad_hip.clip_grad_norm(flat, self.grad_clip)  # rnad.py:456
        else:
            nn.utils.clip_grad_norm_(self.net.parameters(), self.grad_clip)  # rnad.py:456

    # ------------------------------------------------------------------ reference learn/rnad.py:502-523
    def train_step(self, buffer, alpha, log=None):
        """One iteration of the reference's inner loop: rollout (every buffer_mod steps) -> buffer sample -> __learn ->
        Adam -> EMA target.  Also what bench.py times as one "step".

        RNaD.use_graph (default on): on-policy steps of the per-row tabular mode are captured ONCE as a hipGraph (every kernel of
        the step is enqueued on torch's current stream, so torch.cuda.graph records them all) and replayed afterwards: the host
        then only rewrites the 16 bytes of per-step scalars (noise seed, alpha: struct rnad_step_params in device memory) and
        launches the graph -- ~30 launches and their Python bookkeeping become one call.  Same kernels, same inputs: the replayed
        steps are bit-identical to the eager ones (tests/test_hip_graph.py)."""
        # this trainer's own workspaces of the bucketed pipeline (sort scratch, the learner's 64-bit accumulators, staging buffers): a
        # second trainer over the same tree and batch size -- main.py:55-81 builds several -- may step on another stream meanwhile
        with rnad_hip.workspace_owner(self._workspace_token()):
            self._leaf_watch()  # (looks at the PREVIOUS step's batch, before this step's graph key is taken)
            if self._graph_eligible(buffer, log):
                return self._graph_step(buffer, alpha)
            return self._step_body(buffer, alpha, log)

    def _workspace_token(self):
        token = self.__dict__.get("_ws_token")
        if token is None:
            token = self._ws_token = rnad_hip.WorkspaceToken()
        return token

    def _step_body(self, buffer, alpha, log=None, step_params=None):
        world, rank = self._world, self._rank
        local_batch = self.batch_size // world
        handle = self.tree.handle()
        T_cap = 2 * handle.max_depth
        mode = False if self.reuse_actor_outputs else self._tabular_mode(T_cap, local_batch)
        tables = None
        lazy = mode is True and self._use_lazy_rows(handle, local_batch, T_cap, log, buffer)
        visited = None
        fold = mode is True and self._fold()
        if lazy:
            tables = self._table_outputs(alpha, getattr(self, "obs_half", False), policy_only=True, fold=fold)
            visited = torch.empty((2 * handle.S,), dtype=torch.int32, device=self.device)
        elif mode is True:
            # the nets do not change between this step's rollout and its update: one evaluation of the 2S observations serves the
            # actor (= the learner net, rnad.py:503-505) and all four nets of __learn
            on_policy = self._plays_what_it_learns(handle, local_batch, T_cap, buffer)
            shard = self._shard_now(handle, local_batch, log, lazy, on_policy)
            tables = self._table_outputs(alpha, getattr(self, "obs_half", False), want_target_logits=log is not None, fold=fold,
                                         records_hp=self._learn_params(alpha), step_params=step_params, shard=shard,
                                         dedup=self._dedup_now(handle, log, lazy, shard, fold, on_policy),
                                         # (the compact rollout's keys pass carries the copies; any other rollout needs them made first)
                                         defer_expand=on_policy)
        if self.total_steps % self.buffer_mod == 0:
            episodes = episode.Episodes(self.tree, local_batch, seed=self._new_seed(), lane_offset=rank * local_batch,
                                        obs_half=getattr(self, "obs_half", False))
            # no host sync: trailing all-absorbed steps are masked by `valid`
            store_values = self.reuse_actor_outputs or getattr(self, "store_actor_values", False)
            # rollout and learner in one launch (r04, RNaD.fuse_rollout_learner): the batch played here is the batch learned from -- a
            # one-batch buffer refilled every step --, its records exist before the rollout (no lazy rows) and nothing is logged
            learn_now = None
            if (mode is True and not lazy and log is None and self._fuse_now() and self.buffer_mod == 1
                    and getattr(buffer, "max_size", None) == 1 and tables.get("fast_records") is not None and not store_values
                    and getattr(self, "compact_trajectory", True) and float(self.neurd_clip) < 2.0 ** 28):
                plan = rnad_hip.bucket_plan(handle, local_batch)
                dedup = tables.get("dedup")
                grouped = (dedup if (dedup is not None and plan is not None and getattr(self, "group_sums_in_finish", True)
                                     and dedup.groups_below_cut(handle, plan)) else None)
                if plan is not None:
                    # (data parallel: the finish is __learn's, after the all-reduce of the normalisers -- or, on a tree whose normalisers
                    # are known without one, the call's own; row sharding: after an all-reduce of the sums in any case)
                    known = self._known_norm(T_cap) if not shard else None
                    leaf = self._leaf_now(handle, local_batch, T_cap)
                    learn_now = dict(fast_records=tables["fast_records"], hp=self._learn_params(alpha), norm_is_global=not self._dp(),
                                     norm_global=known, leaf=leaf,
                                     distinct=self._distinct_now() and leaf is None,
                                     rows=grouped.singles if grouped is not None else tables.get("rows"), groups=grouped)
            episodes.generate(self.net, trim=False, keep_logits=self.reuse_actor_outputs,
                              skip_absorbed=getattr(self, "skip_absorbed", True) and not self.reuse_actor_outputs,
                              store_values=store_values, tabular=bool(mode), bucketed=mode is True,
                              logits_table=tables["logit"] if tables is not None else None,
                              value_table=tables["v"] if tables is not None and store_values else None,
                              policy_table=((tables["records"], rnad_hip.policy_column(self.tree.max_actions))
                                            if tables is not None and not lazy else None),
                              step_params=step_params, compact=getattr(self, "compact_trajectory", True), visited=visited,
                              # single process: the learner's launch adds up the alive counts (one kernel less); data parallel: the
                              # normalisers are all-reduced beside the learner kernel, so they are needed before it
                              defer_alive=mode is True and log is None and not self._dp(),
                              staged_actor=tables.get("staged_actor") if (tables is not None and lazy) else None, learn=learn_now)
            if tables is not None and getattr(tables.get("records"), "_expand", None) is not None:
                # (the rollout took a path that does not carry the copies of the distinct-observation tables: make them now)
                rnad_hip.rows_expand(*tables["records"]._expand)
                tables["records"]._expand = None
                rnad_hip.complete_records(tables["records"])
            if lazy:
                # the rows this batch went through are known now: value heads, records, gradient tables, backward on those only
                assert episodes._compact is not None, "lazy rows need the compact bucketed rollout"
                self._value_tables(tables, visited, alpha, step_params=step_params)
                episodes._compact = (episodes._compact[0], tables["records"])
            episodes._actor_tag = (id(self.net), self.total_steps)
            buffer.append(episodes)
            self.last_episodes = episodes
        episodes_sample = buffer.sample(local_batch)
        fused_tail = self._fused_tail()
        self.__learn(episodes_sample, alpha, log=log, tables=tables, defer_clip=fused_tail is not None)
        if fused_tail is not None and self._pending_flat is not None:
            # clip + Adam + EMA target in one launch (csrc/optim.hip); a captured step also moves its queue of per-step scalars on
            fused_tail(self._pending_flat, advance=step_params)
            self._tail_advances = step_params is not None
            self._pending_flat = None
            self.optimizer.zero_grad()
            return
        if self._pending_flat is not None:
            rnad_hip.clip_grad_norm(self._pending_flat, self.grad_clip)
            self._pending_flat = None
        self.optimizer.step()
        self.optimizer.zero_grad()
        # EMA target, rnad.py:516-523: target <- gamma * net + (1 - gamma) * target for every state_dict entry, as two
        # multi-tensor kernels instead of 3 launches per entry
        with torch.no_grad():
            tgt = [t for t in self.net_target.state_dict().values() if t.is_floating_point()]
            src = [t for k, t in self.net.state_dict().items() if t.is_floating_point()]
            torch._foreach_mul_(tgt, 1 - self.gamma_averaging)
            torch._foreach_add_(tgt, src, alpha=self.gamma_averaging)

    def _fused_tail(self):
        """rnad_hip.OptimizerStep over the learner's eight tensors (clip + Adam + EMA in one launch), or None when the plain torch
        sequence must run: another net type, a CPU run, an optimiser that is not the reference's Adam (rnad.py:232-237), or Adam
        state that does not exist yet (torch creates it in its first step())."""
        if not getattr(self, "fused_optimizer", True) or not self._fused_mlp() or not isinstance(self.net_target, net.MLP):
            return None
        opt = self.optimizer
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return None
        grp = opt.param_groups[0]
        if grp.get("amsgrad") or grp.get("weight_decay") or grp.get("maximize") or grp.get("differentiable"):
            return None
        weights = self.net._weights()
        fold = bool(getattr(self, "_layout_in_use", False))
        images = self._packed_images(fold)  # (the layout this step's table evaluations used: the tail keeps exactly these two buffers current)
        key = (id(opt), id(self.net), id(self.net_target), grp["lr"], tuple(grp["betas"]), grp["eps"], self.grad_clip, self.gamma_averaging,
               fold, images[0].data_ptr())
        cached = getattr(self, "_fused_tail_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        states = [opt.state.get(w) for w in weights]
        if any(not st or "exp_avg" not in st for st in states):
            return None
        steps = [st["step"] for st in states]
        if any((not torch.is_tensor(x)) or (not x.is_cuda) or x.dtype != torch.float32 for x in steps):
            return None
        lr = grp["lr"]
        if torch.is_tensor(lr):
            return None
        tail = rnad_hip.OptimizerStep(weights, [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states], steps,
                                      self.net_target._weights(), lr, grp["betas"][0], grp["betas"][1], grp["eps"], self.grad_clip,
                                      self.gamma_averaging, packed=tuple(images), A=self.tree.max_actions, fold=fold)
        self._fused_tail_cache = (key, tail)
        self._packed_cache["maintained"] = fold
        return tail

    # ------------------------------------------------------------------ hipGraph replay of the on-policy tabular step
    _GRAPH_WARMUP = 3  # eager steps before capture: code objects loaded, allocator pools and the table caches exist

    def _graph_eligible(self, buffer, log):
        if not getattr(self, "use_graph", True) or log is not None or self.reuse_actor_outputs:
            return False
        if self._dp() and (dist.get_backend() != "nccl" or os.environ.get("RNAD_GRAPH_DIST", "1") == "0"):
            return False  # RCCL collectives are captured with the step (torch's NCCL backend supports stream capture); gloo cannot be
        dev = self.device if isinstance(self.device, torch.device) else torch.device(self.device)
        if dev.type != "cuda" or self.buffer_mod != 1 or buffer.max_size != 1:
            return False
        g = getattr(self, "_graph", None)
        if g is not None and g.get("failed"):
            return False
        handle = self.tree.handle()
        local_batch = self.batch_size // self._world
        if self._tabular_mode(2 * handle.max_depth, local_batch) is not True or rnad_hip.bucket_plan(handle, local_batch) is None:
            return False
        if not all(group.get("capturable", False) for group in self.optimizer.param_groups):
            return False
        return True

    def _graph_key(self, buffer):
        """Everything a captured step has baked in: replay is only valid while none of it changed."""
        grp = self.optimizer.param_groups[0]
        lr = grp["lr"]
        opt = (float(lr) if not torch.is_tensor(lr) else id(lr), tuple(float(b) for b in grp["betas"]), float(grp["eps"]))  # by value in the captured launch
        return (id(buffer), id(self.net), id(self.net_target), id(self.net_reg), id(self.net_reg_), id(self.optimizer), id(self.tree.handle()), opt,
                self.batch_size, self.tabular, getattr(self, "tabular_gate", 8), self.eta, self.beta, self.neurd_clip, self.grad_clip,
                self.c_bar, self.roh_bar, self.vtrace_gamma, self.value_weight, self.neurd_weight, self.epsilon_threshold, self.n_discrete,
                self.gamma_averaging, getattr(self, "obs_half", False), getattr(self, "store_actor_values", False), getattr(self, "fused_optimizer", True),
                getattr(self, "compact_trajectory", True), getattr(self, "lazy_rows", None), rnad_hip.plan_knobs(), os.environ.get("RNAD_LEAF_CHUNK"),
                getattr(self, "fold_legal", True), self._fuse_now(), self._fuse_now() and self._distinct_now(), getattr(self, "analytic_norm", True), os.environ.get("RNAD_FUSED_DISTINCT"), os.environ.get("RNAD_FUSED_CHUNK"),
                getattr(self, "leaf_paths", None), os.environ.get("RNAD_LEAF_PATHS"), os.environ.get("RNAD_LEAF_CROWDED_LANES"))

    def _graph_step(self, buffer, alpha):
        g = getattr(self, "_graph", None)
        key = self._graph_key(buffer)
        if g is None or g["key"] != key:
            g = self._graph = {"key": key, "eager_steps": 0, "graph": None, "failed": False}
        if g["graph"] is None and g["eager_steps"] < self._GRAPH_WARMUP:
            g["eager_steps"] += 1
            return self._step_body(buffer, alpha, None)
        seed = self._new_seed()
        if g["graph"] is None:
            g["dev"] = torch.zeros((rnad_hip.STEP_QUEUE_WORDS,), dtype=torch.int64, device=self.device)
            g["ahead"], g["pos"], g["advances"], g["queue_sets"] = None, 0, False, 0
        # The step's scalars (noise seed, alpha) are in device memory.  The captured optimiser launch moves a queue of them on at the end
        # of every step (rnad_hip.step_queue_set), so a replay needs no launch of its own as long as this step's scalars are the ones
        # queued for it: the seeds are a counter hash, the alphas come from the caller's schedule (alpha_ahead; else "as now").  Anything
        # else -- a logging step in between took a seed, alpha left the prediction, the queue ran out -- sets the queue again.
        entry = rnad_hip.step_entry(seed, alpha)
        if not (g["ahead"] is not None and g["pos"] < len(g["ahead"]) and g["ahead"][g["pos"]] == entry):
            ahead = getattr(self, "alpha_ahead", None)
            n = rnad_hip.STEP_QUEUE if (g["advances"] or g["graph"] is None) else 1
            g["ahead"] = [entry] + [rnad_hip.step_entry(self._seed_at(self._seed_count + k), alpha if ahead is None else ahead(k))
                                    for k in range(1, n)]
            g["pos"] = 0
            g["queue_sets"] += 1
            rnad_hip.step_queue_set(g["dev"], g["ahead"])
        # the regularisation nets are constant inside a captured step: refresh their tables (in place) when their weights changed
        self._reg_tables(self.tree.handle().observations_table(getattr(self, "obs_half", False)), self._fold())
        self._packed_images()  # (re-packed here, in place, if somebody edited net / net_target since the last step: no pack inside the graph)
        if g["graph"] is None:
            graph = torch.cuda.CUDAGraph()
            self._seed_override = seed  # the captured body (and an eager retry) must use THIS step's seed, not draw another
            self._tail_advances = False
            ok = False
            import gc

            gc_was_on = gc.isenabled()
            gc.disable()  # (a collection in the middle of the capture may run finalisers that free device memory: hipFree invalidates it)
            try:
                # (data parallel: torch's NCCL watchdog thread polls the events of the eager steps' collectives; under the default GLOBAL
                # capture mode such a query from another thread while this one captures is an error -- seen once in a full test run as a
                # crash of the watchdog.  thread_local: only this thread's calls are checked)
                with torch.cuda.graph(graph, capture_error_mode="thread_local" if self._dp() else "global"):
                    self._step_body(buffer, alpha, None, step_params=g["dev"])
                # only the native bucketed rollout takes its seed from device memory: anything else would replay stale noise
                ok = getattr(self.last_episodes, "buckets", None) is not None
                why = "the rollout of this step is not the native bucketed one"
            except Exception as err:  # capture is an optimisation: fall back to eager steps, loudly
                why = str(err)
            finally:
                if gc_was_on:
                    gc.enable()
            if self._dp():  # every rank replays, or none does (a rank replaying collectives the others enqueue eagerly would hang)
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if ok and int(flag.item()) == 0:
                    ok, why = False, "another rank could not capture its step"
            if not ok:
                logging.warning("hipGraph capture of the training step not used (%s); continuing with eager steps", why)
                g["failed"] = True
                del graph
                torch.cuda.synchronize()
                try:
                    return self._step_body(buffer, alpha, None)
                finally:
                    self._seed_override = None
            self._seed_override = None
            g["graph"] = graph
            g["episodes"] = self.last_episodes
            g["advances"] = self._tail_advances  # (the captured step ends in the fused optimiser launch)
        g["graph"].replay()
        rnad_hip.destroy_deferred()  # (tree handles that died while the capture was open: freed here, outside any capture)
        if g["advances"]:
            g["pos"] += 1
        else:
            g["ahead"] = None
        # the replay rewrote the CAPTURED batch in place: an eager (logging) step in between left another Episodes object in
        # last_episodes and in the buffer
        ep = g["episodes"]
        if self.last_episodes is not ep:
            self.last_episodes = ep
            buffer.append(ep)
        ep.seed = ep.states.seed = seed
        ep.invalidate_derived()

    def initialize(self):
        """Public alias of the reference's private __initialize (nets, optimizer, checkpoint 0/0)."""
        self.__initialize()

    # ------------------------------------------------------------------ reference learn/rnad.py:458-531
    def __resume(self, max_updates=10**6, checkpoint_mod=1000, expl_mod=1, log_mod=20) -> None:
        buffer = episode.Buffer(self.n_batches_per_buffer)
        rank = self._rank
        assert self.batch_size % self._world == 0, "batch_size must be divisible by the number of GPUs"
        for _ in range(max_updates):
            delta_m = self._steps_of_current_update()
            if delta_m is None:
                return
            logging.info("m: {}, delta_m: {}".format(self.m, delta_m))
            buffer.max_size = self.n_batches_per_buffer

            if self.m % expl_mod == 0 and self.n == 0 and self.m != 0:
                if rank == 0:
                    nashconv = self._evaluate_nashconv()
                    self.nashconv_history.append((self.m, self.total_steps, nashconv))
                    if self.wandb:
                        import wandb

                        wandb.log({"nashconv": nashconv}, step=self.total_steps)
                if self._dp():
                    dist.barrier()

            while self.n < delta_m:
                alpha = self.alpha_of(self.n, delta_m)
                self.alpha_ahead = lambda k, n=self.n, d=delta_m: self.alpha_of(n + k, d)  # (what the next steps will ask for: _graph_step)
                if self.n % checkpoint_mod == 0:
                    self.__save_checkpoint()
                # the reference logs only with wandb on (rnad.py:509); keep_last_log asks for the same scalars in RNaD.last_log
                log = {} if (self.n % log_mod == 0 and (self.wandb or self.keep_last_log)) else None
                self.train_step(buffer, alpha, log=log)
                if log:
                    self.last_log = dict(log, m=self.m, n=self.n, total_steps=self.total_steps)
                    if self.wandb:
                        import wandb

                        wandb.log(log, step=self.total_steps)
                self.n += 1
                self.total_steps += 1

            self.n = 0
            self.m += 1
            self.net_reg_.load_state_dict(self.net_reg.state_dict())  # rnad.py:530-531
            self.net_reg.load_state_dict(self.net_target.state_dict())

    def run(self, max_updates=10**6, checkpoint_mod=1000, expl_mod=1, log_mod=20):
        """Either starts or resumes a run (reference learn/rnad.py:533-547)."""
        self.__initialize()
        self.__resume(max_updates=max_updates, checkpoint_mod=checkpoint_mod, expl_mod=expl_mod, log_mod=log_mod)
        if self.wandb:
            import wandb

            wandb.finish()
