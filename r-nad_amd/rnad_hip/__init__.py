"""ctypes binding of librnad_hip.so (include/rnad_hip.h) for torch device tensors.

Thin plumbing only: every function checks dtype/device/contiguity, passes `tensor.data_ptr()` and
torch's CURRENT stream to the C-ABI and raises `RnadHipError` on a non-zero status.  There is no
CPU fallback: if the library is missing or a tensor is not on a GPU the call fails loudly.
"""
import ctypes as C
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.realpath(__file__))
SO_PATH = os.environ.get("RNAD_HIP_SO") or os.path.join(_HERE, "..", "csrc", "librnad_hip.so")  # override: kernel A/B experiments

MAX_ACTIONS = 8
MAX_TRANSITIONS = 8


class RnadHipError(RuntimeError):
    pass


class Traj(C.Structure):
    """struct rnad_traj (include/rnad_hip.h)."""

    _fields_ = [
        ("T_cap", C.c_int32), ("obs_half", C.c_int32), ("B", C.c_int64),
        ("indices", C.c_void_p), ("observations", C.c_void_p), ("mask_bits", C.c_void_p),
        ("policy", C.c_void_p), ("actions", C.c_void_p), ("rewards", C.c_void_p),
        ("values", C.c_void_p), ("alive", C.c_void_p),
    ]


class LearnParams(C.Structure):
    """struct rnad_learn_params (include/rnad_hip.h)."""

    _fields_ = [
        ("alpha", C.c_float), ("one_minus_alpha", C.c_float),
        ("eta", C.c_float), ("lambda_", C.c_float), ("c", C.c_float), ("rho", C.c_float), ("gamma", C.c_float),
        ("clip", C.c_float), ("threshold", C.c_float), ("w_v", C.c_float), ("w_n", C.c_float),
        ("eps_threshold", C.c_float), ("n_disc", C.c_int32),
    ]


_lib = None
HEADER_PATH = os.path.join(_HERE, "..", "..", "include", "rnad_hip.h")

_SCALARS = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float,
            "double": C.c_double}


def _ctype_of(decl, what):
    """C parameter / return declaration -> ctypes type.  Every pointer travels as c_void_p (tensor.data_ptr(), byref(struct),
    arrays of pointers, or None)."""
    decl = decl.strip()
    if "*" in decl:
        return C.c_char_p if what == "return" and "char" in decl else C.c_void_p
    words = [w for w in decl.replace("const", " ").split()]
    base = words[0] if what == "return" else (words[-2] if len(words) > 1 else words[-1])
    if base == "void":
        return None
    if base not in _SCALARS:
        raise RnadHipError(f"include/rnad_hip.h: cannot map '{decl}' to a ctypes type")
    return _SCALARS[base]


def header_prototypes(path=HEADER_PATH):
    """{name: (restype, [argtypes])} for every function include/rnad_hip.h declares, so that the binding can never disagree with
    the ABI about an argument's width (a bare Python int would otherwise travel as a C int)."""
    import re

    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(rnad_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        args = [] if params in ("", "void") else [_ctype_of(a, "param") for a in params.split(",")]
        protos[name] = (_ctype_of(ret, "return"), args)
    return protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RnadHipError(
                f"{os.path.normpath(SO_PATH)} is missing: build it with `make -C r-nad_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback for the R-NaD hot path."
            )
        handle = C.CDLL(SO_PATH)
        for name, (restype, argtypes) in header_prototypes().items():
            fn = getattr(handle, name)  # AttributeError: the library is older than the header
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    return _lib


def _check(rc):
    if rc != 0:
        raise RnadHipError(lib().rnad_last_error().decode() or f"librnad_hip status {rc}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dp(t, dtype, name, optional=False):
    """Device pointer of a contiguous CUDA tensor of the given dtype."""
    if t is None:
        if optional:
            return None
        raise RnadHipError(f"{name}: tensor required")
    if not t.is_cuda:
        raise RnadHipError(f"{name}: expected a GPU tensor, got device {t.device} (no CPU path)")
    if t.get_device() != torch._C._cuda_getDevice():
        # kernels are launched on HIP's current device and on torch's current stream of that device
        raise RnadHipError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device (or use `with torch.cuda.device(...)`) first")
    if t.dtype != dtype:
        raise RnadHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RnadHipError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


F32, F64, I32, U8, F16 = torch.float32, torch.float64, torch.int32, torch.uint8, torch.float16


# --------------------------------------------------------------------------------------- tree
class TreeHandle:
    """Owns an `rnad_tree_t*`: the packed node / transition tables on one GPU."""

    def __init__(self, index, value, chance, expected_value, legal, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RnadHipError(f"tree tables can only be uploaded to a GPU, got device {dev}")
        host = lambda t, dt: t.detach().to("cpu", dt).contiguous()  # noqa: E731
        index, value, chance = host(index, torch.int64), host(value, F32), host(chance, F32)
        expected_value, legal = host(expected_value, F32), host(legal, F32)
        S, Cc, A, A2 = index.shape
        assert A == A2 and value.shape == index.shape and chance.shape == index.shape
        assert expected_value.shape == (S, 1, A, A) and legal.shape == (S, 1, A, A)
        self.S, self.C, self.A = S, Cc, A
        self.device = dev
        self._h = C.c_void_p()
        _destroy_deferred()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        _check(lib().rnad_tree_create(C.byref(self._h), C.c_int64(S), Cc, A, C.c_void_p(index.data_ptr()),
                                      C.c_void_p(value.data_ptr()), C.c_void_p(chance.data_ptr()),
                                      C.c_void_p(expected_value.data_ptr()), C.c_void_p(legal.data_ptr()), idx))
        self.max_depth = int(lib().rnad_tree_info(self._h, 3))
        self.table_bytes = int(lib().rnad_tree_info(self._h, 5))
        self.uniform_length = bool(lib().rnad_tree_info(self._h, 6))  # every episode 2 * max_depth env steps long

    @property
    def ptr(self):
        return self._h

    def observations_table(self, half=False):
        """[2S, 2, A, A]: the observation of every (player to move, state), row = player * S + state (rnad_observe_all).
        Built on first use and kept; half=True rounds it to fp16 (what an fp16 trajectory buffer would hold)."""
        key = "_obs_table_half" if half else "_obs_table"
        if getattr(self, key, None) is None:
            row, col = observe_all(self, self.device)
            tab = torch.cat([row, col], dim=0)
            setattr(self, key, tab.half() if half else tab)
            _publish_shared(tab.device)
        return getattr(self, key)

    def is_foldable_table(self, obs):
        """True for this tree's own observation table (fp32 or fp16) when its legal planes allow the FOLD kernels: the only inputs the
        binding hands to them -- they read legal[0][1] of a row and assume the rest of the plane."""
        return self.legal_foldable and any(obs is getattr(self, k, None) for k in ("_obs_table", "_obs_table_half"))

    @property
    def legal_foldable(self):
        """True when every (player, state) row of the observation table carries an all-ones legal plane, or -- the absorbing state -- e0 =
        [1, 0, ..., 0]: the premise of the FOLD instantiations of the MLP kernels (include/rnad_hip.h "The legal fold").  Checked once."""
        if getattr(self, "_legal_foldable", None) is None:
            ok = False
            if self.A >= 2:
                legal = self.observations_table()[:, 1].reshape(2 * self.S, -1)
                e0 = torch.zeros_like(legal[0])
                e0[0] = 1.0
                ones = (legal == 1.0).all(dim=1)
                ok = bool((ones | (legal == e0).all(dim=1)).all().item())
            self._legal_foldable = ok
        return self._legal_foldable

    def obs_dedup(self, half=False):
        """The rows of the observation table grouped by the BITS of their observation (csrc/rows_dedup.hip).  Returns an object with
        n_rows (2S), n_unique, uniq (RowList of one representative row per group, ascending: the first row of the group), rep_of (int32
        [2S]: the representative of every row), multi_start / multi_order (int32 CSR of the groups with more than one row, rows ascending,
        the representative first) and n_multi.  Computed once per table and kept."""
        key = "_obs_dedup_half" if half else "_obs_dedup"
        got = getattr(self, key, None)
        if got is None:
            table = self.observations_table(half)
            N = table.shape[0]
            flat = table.reshape(N, -1)
            bits = flat.view(torch.int16 if half else torch.int32).to(torch.int64)
            _, inverse = torch.unique(bits, dim=0, return_inverse=True)
            n_unique = int(inverse.max().item()) + 1
            rows = torch.arange(N, device=table.device, dtype=torch.int64)
            first = torch.full((n_unique,), N, dtype=torch.int64, device=table.device).scatter_reduce(0, inverse, rows, reduce="amin")
            rep_of = first[inverse]
            uniq_rows = torch.sort(first).values
            # groups with more than one row: their rows, ascending, group after group in the order of their representatives
            size = torch.bincount(inverse, minlength=n_unique)
            in_multi = size[inverse] > 1
            m_rows = rows[in_multi]
            order = m_rows[torch.argsort(rep_of[in_multi] * N + m_rows)]
            reps_multi = torch.sort(first[size > 1]).values
            counts = size[inverse[reps_multi]]
            start = torch.zeros((reps_multi.numel() + 1,), dtype=torch.int64, device=table.device)
            start[1:] = torch.cumsum(counts, 0)
            got = ObsDedup()
            got.n_rows, got.n_unique, got.n_multi = N, n_unique, int(reps_multi.numel())
            got.singles = RowList(rows[~in_multi].to(I32), N, table.device)  # the rows that are groups of their own, ascending
            got._in_multi = in_multi
            got.uniq = RowList(uniq_rows.to(I32), N, table.device)
            got.rep_of = rep_of.to(I32).contiguous()
            got.multi_start = start.to(I32).contiguous()
            got.multi_order = order.to(I32).contiguous()
            # the groups' first 256 rows once more, padded per group ([n_multi][4][64], -1 beyond the group): rnad_row_groups_t.first
            pos = start[:-1, None] + torch.arange(256, device=table.device, dtype=torch.int64)[None, :]
            inside = pos < start[1:, None]
            got.multi_first = torch.where(inside, order[pos.clamp(max=max(order.numel() - 1, 0))] if order.numel() else pos, -torch.ones_like(pos)).to(I32).contiguous()
            got.c_groups = RowGroups(got.n_multi, got.multi_start.data_ptr(), got.multi_order.data_ptr(), got.multi_first.data_ptr())
            _publish_shared(table.device)
            setattr(self, key, got)
        return got

    def __del__(self):
        # rnad_tree_destroy is a series of hipFree calls, and a hipFree while ANY stream of the process is capturing (torch.cuda.graph
        # captures in the global mode) invalidates that capture -- which is what happens when the garbage collector gets to the tree of an
        # earlier trainer in the middle of RNaD's capture of a step.  Handles that die during a capture are parked and destroyed at the
        # next safe point (the next handle's construction or destruction outside a capture).
        try:
            if self._h:
                h, self._h = self._h, C.c_void_p()
                _deferred_destroy.append(h)
                _destroy_deferred()
        except Exception:
            pass


def _publish_shared(device):
    """A cache kept on a tree handle is read by every user of the handle, possibly on other streams (two trainers of one tree on two
    streams): whatever built it on the current stream is complete before anyone else can see it.  Not during a stream capture (a
    synchronize would invalidate it; a captured step only ever meets caches its eager warm-up steps have built)."""
    if device is None or torch.device(device).type != "cuda":
        return
    if torch.cuda.is_current_stream_capturing():
        return
    torch.cuda.current_stream(device).synchronize()


class RowGroups(C.Structure):
    """struct rnad_row_groups (include/rnad_hip.h)."""

    _fields_ = [("n_groups", C.c_int32), ("start", C.c_void_p), ("order", C.c_void_p), ("first", C.c_void_p), ("rows_below_cut", C.c_int32)]


class ObsDedup:
    """TreeHandle.obs_dedup's result (fields there)."""

    def groups_below_cut(self, tree, plan):
        """May k_bucket_finish add the groups up (the `groups` of bucket_finish / learn_bucketed_compact)?  The sums of the rows above the
        buckets travel through the replicas, so each of those rows must be a group of its own.  Checked once per batch size."""
        known = self.__dict__.setdefault("_below_cut", {})
        if plan.B not in known:
            bucket_of, n_groups = bucket_map(tree, plan.B)
            upper = torch.nonzero(bucket_of >= n_groups).reshape(-1).to(self._in_multi.device)
            known[plan.B] = not bool(self._in_multi[torch.cat([upper, upper + tree.S])].any().item())
        return known[plan.B]


_deferred_destroy = []


def _destroy_deferred():
    try:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
    except Exception:
        return
    while _deferred_destroy:
        lib().rnad_tree_destroy(_deferred_destroy.pop())


def destroy_deferred():
    """Free the tables of tree handles that died while a stream capture was open (a hipFree would have invalidated the capture, so they
    were parked).  Called at safe points: handle construction / destruction, bucket_plan, after RNaD's graph replays."""
    if _deferred_destroy:
        _destroy_deferred()


def observe(tree, idx, player, obs=None, half=False, mask_bits=None, mask=None):
    """K1.  idx int32 [B] -> obs [B,2,A,A] (fp32 or fp16), optional mask_bits u8 [B], mask f32 [B,A]."""
    B, A = idx.shape[0], tree.A
    if obs is None:
        obs = torch.empty((B, 2, A, A), dtype=F16 if half else F32, device=idx.device)
    if B == 0:
        return obs
    _check(lib().rnad_observe(tree.ptr, C.c_int64(B), _dp(idx, I32, "idx"), int(player), _dp(obs, F16 if half else F32, "obs"),
                              int(half), _dp(mask_bits, U8, "mask_bits", True), _dp(mask, F32, "mask", True), _stream()))
    return obs


def observe_all(tree, device):
    S, A = tree.S, tree.A
    row = torch.empty((S, 2, A, A), dtype=F32, device=device)
    col = torch.empty_like(row)
    _check(lib().rnad_observe_all(tree.ptr, _dp(row, F32, "obs_row"), _dp(col, F32, "obs_col"), _stream()))
    return row, col


# --------------------------------------------------------------------------------------- policy head / sampler / transition
def policy_head(logits, mask_bits=None, mask=None, want_log=False):
    A = logits.shape[-1]
    N = logits.numel() // A
    policy = torch.empty_like(logits)
    log_policy = torch.empty_like(logits) if want_log else None
    _check(lib().rnad_policy_head(C.c_int64(N), A, _dp(logits, F32, "logits"), _dp(mask_bits, U8, "mask_bits", True),
                                  _dp(mask, F32, "mask", True), _dp(policy, F32, "policy"),
                                  _dp(log_policy, F32, "log_policy", True), _stream()))
    return (policy, log_policy) if want_log else policy


def sample(probs, noise=None, seed=0, lane0=0, step=0, stream_id=0):
    B, n = probs.shape
    out = torch.empty((B,), dtype=I32, device=probs.device)
    _check(lib().rnad_sample(C.c_int64(B), n, _dp(probs, F32, "probs"), _dp(noise, F32, "noise", True), C.c_uint64(seed),
                             C.c_int64(lane0), int(step), int(stream_id), _dp(out, I32, "out"), _stream()))
    return out


def transition(tree, idx, row_actions, col_actions, noise=None, seed=0, lane0=0, step=1, alive=None):
    B = idx.shape[0]
    idx_out = torch.empty_like(idx)
    reward = torch.empty((B,), dtype=F32, device=idx.device)
    _check(lib().rnad_transition(tree.ptr, C.c_int64(B), _dp(idx, I32, "idx"), _dp(row_actions, I32, "row_actions"),
                                 _dp(col_actions, I32, "col_actions"), _dp(noise, F32, "noise", True), C.c_uint64(seed),
                                 C.c_int64(lane0), int(step), _dp(idx_out, I32, "idx_out"), _dp(reward, F32, "reward"),
                                 _dp(alive, I32, "alive", True), _stream()))
    return idx_out, reward


# --------------------------------------------------------------------------------------- fused MLP
MLP_KEYS = ("value_fc0.weight", "value_fc0.bias", "value_fc1.weight", "value_fc1.bias",
            "policy_fc0.weight", "policy_fc0.bias", "policy_fc1.weight", "policy_fc1.bias")


def mlp_pack(weights, A):
    """The eight Linear tensors (MLP_KEYS order, fp32, device) -> the packed LDS image the MLP kernels read."""
    W = weights[0].shape[0]
    packed = torch.empty((lib().rnad_mlp_packed_size(A, W),), dtype=F32, device=weights[0].device)
    _check(lib().rnad_mlp_pack(A, W, *[_dp(w.detach(), F32, "weight") for w in weights], _dp(packed, F32, "packed"), _stream()))
    return packed


def mlp_pack_many(weight_lists, A, out=None, fold=False):
    """mlp_pack for up to four nets of one shape in ONE launch (rnad_mlp_pack_multi) -> list of packed images (written into `out`,
    a list of preallocated images, when given).  fold: the images of the FOLD kernels (rnad_mlp_pack_fold_multi)."""
    n = len(weight_lists)
    assert 1 <= n <= 4 and all(len(w) == 8 for w in weight_lists)
    W = weight_lists[0][0].shape[0]
    size = mlp_packed_size(A, W, fold)
    dev = weight_lists[0][0].device
    outs = [torch.empty((size,), dtype=F32, device=dev) for _ in range(n)] if out is None else list(out)
    assert len(outs) == n and all(o.numel() == size for o in outs)
    wp = (C.c_void_p * (8 * n))(*[_dp(w.detach(), F32, "weight").value for ws in weight_lists for w in ws])
    op = (C.c_void_p * n)(*[_dp(o, F32, "packed").value for o in outs])
    _check((lib().rnad_mlp_pack_fold_multi if fold else lib().rnad_mlp_pack_multi)(n, A, W, wp, op, _stream()))
    return outs


def mlp_packed_size(A, W, fold=False):
    return int((lib().rnad_mlp_fold_packed_size if fold else lib().rnad_mlp_packed_size)(A, W))


class LiveRows:
    """Ascending list of the positions with indices != 0 (rnad_compact_valid): `rows` int32 [N] of which the first `count`
    (a device int64) are meaningful.  Built and consumed on the stream, never read by the host."""

    def __init__(self, indices):
        flat = indices.reshape(-1)
        N = flat.numel()
        dev = flat.device
        self.N = N
        self.rows = torch.empty((N,), dtype=I32, device=dev)
        self.count = torch.empty((1,), dtype=torch.int64, device=dev)
        self._scratch = torch.empty((max(int(lib().rnad_compact_workspace(C.c_int64(N))), 1),), dtype=I32, device=dev)
        _check(lib().rnad_compact_valid(C.c_int64(N), _dp(flat, I32, "indices"), _dp(self.rows, I32, "rows"),
                                        C.c_void_p(self.count.data_ptr()), _dp(self._scratch, I32, "block_counts"), _stream()))


def compact_valid(indices):
    """indices int32 [...] -> LiveRows of the flattened positions that are not in the absorbing state."""
    return LiveRows(indices)


class RowList:
    """A fixed list of rows in the shape of a LiveRows (rows int32 [n], count int64 [1] on the device), e.g. the upper rows of a cut."""

    def __init__(self, rows, N, device):
        self.N = N
        self.rows = torch.as_tensor(rows, dtype=I32).to(device).contiguous()
        self.count = torch.tensor([self.rows.numel()], dtype=torch.int64, device=device)


def _fold_checked(fold, obs, who):
    """fold: False, or the TreeHandle whose observation table `obs` is (the FOLD kernels silently assume every row's legal plane is all
    ones or e0: that is a property of a tree's table, checked once by TreeHandle.legal_foldable, not of arbitrary observations)."""
    if not fold:
        return False
    if not isinstance(fold, TreeHandle) or not fold.is_foldable_table(obs):
        raise RnadHipError(f"{who}: fold= takes the TreeHandle whose observations_table() is `obs`, and that table must be legal_foldable "
                           "(the FOLD kernels read legal[0][1] of a row only)")
    return True


def mlp_forward(packed, W, obs, A, want_logits=True, want_value=True, live=None, out=None, zero_rest=True, fold=False):
    """packed: mlp_pack(weights, A); obs [N, 2, A, A] fp32/fp16 -> logits [N, A], value [N, 1].
    fold: False, or the TreeHandle whose observation table `obs` is (FOLD kernels; packed = mlp_pack_many(..., fold=True)).
    A head that is not wanted is not computed (returns None for it).
    live: a LiveRows / RowList over the N samples -- only those rows are evaluated; the others come back as zeros (zero_rest=False:
    uninitialised -- for callers that never read them).  out = (logits, value): existing tables to write into (their other rows
    are left alone)."""
    fold = _fold_checked(fold, obs, "mlp_forward")
    N = obs.numel() // (2 * A * A)
    half = obs.dtype == F16
    alloc = torch.zeros if (live is not None and zero_rest) else torch.empty
    if out is not None:
        logits, value = out
        want_logits, want_value = logits is not None, value is not None
        assert (logits is None or logits.shape == (N, A)) and (value is None or value.shape == (N, 1))
    else:
        logits = alloc((N, A), dtype=F32, device=obs.device) if want_logits else None
        value = alloc((N, 1), dtype=F32, device=obs.device) if want_value else None
    if fold:  # the FOLD instantiation (packed: mlp_pack_many(..., fold=True); obs rows with a foldable legal plane)
        assert live is None or live.N == N, "live-row list built for a different batch"
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        P = (C.c_void_p * 1)(_dp(packed, F32, "packed").value)
        L = (C.c_void_p * 1)(ptr(logits))
        V = (C.c_void_p * 1)(ptr(value))
        _check(lib().rnad_mlp_forward_fold(1, C.c_int64(N), *_row_list(live), A, W, P, _dp(obs, F16 if half else F32, "obs"), int(half), L, V,
                                           _stream()))
    elif live is None:
        _check(lib().rnad_mlp_forward(C.c_int64(N), A, W, _dp(packed, F32, "packed"), _dp(obs, F16 if half else F32, "obs"), int(half),
                                      _dp(logits, F32, "logits", True), _dp(value, F32, "value", True), _stream()))
    else:
        assert live.N == N, "live-row list built for a different batch"
        _check(lib().rnad_mlp_forward_rows(C.c_int64(N), _dp(live.rows, I32, "rows"), C.c_void_p(live.count.data_ptr()), A, W,
                                           _dp(packed, F32, "packed"), _dp(obs, F16 if half else F32, "obs"), int(half),
                                           _dp(logits, F32, "logits", True), _dp(value, F32, "value", True), _stream()))
    return logits, value


def mlp_forward_actor(tree, packed, W, obs, logits, policy_rows, rows=None, fold=False):
    """rnad_mlp_forward_actor: the policy head of one net on (the listed rows of) the tree's observation table -> `logits` [2S, A] and, from
    the kernel's epilogue, `policy_rows` [2S, policy row stride] -- the actor table of bucket_sort / bucket_play(table_is_policy=True)."""
    fold = _fold_checked(fold, obs, "mlp_forward_actor")
    half = obs.dtype == F16
    assert logits.shape == (2 * tree.S, tree.A) and policy_rows.shape[0] == 2 * tree.S
    assert rows is None or rows.N == 2 * tree.S
    if os.environ.get("RNAD_ROWS_ACTOR", "0") == "1" and lib().rnad_mlp_rows_actor_supported(tree.A, W, int(fold)):
        # (csrc/mlp_rows.hip's mapping for the staged actor: no 112 KB weight image per workgroup at A = 5.  Measured on configs[3]: 3
        # launches 86.9 us against 80.5 for k_mlp_forward -- both spend the same fp32 ALU time -- so it is opt-in)
        _check(lib().rnad_mlp_rows_actor(tree.ptr, *_row_list(rows), W, int(fold), _dp(packed, F32, "packed"), _dp(obs, F16 if half else F32, "obs"),
                                         int(half), _dp(logits, F32, "logits"), _dp(policy_rows, F32, "policy_rows"), _stream()))
        return
    _check(lib().rnad_mlp_forward_actor(tree.ptr, *_row_list(rows), W, int(fold), _dp(packed, F32, "packed"), _dp(obs, F16 if half else F32, "obs"),
                                        int(half), _dp(logits, F32, "logits"), _dp(policy_rows, F32, "policy_rows"), _stream()))


def mlp_forward_multi(packed_list, W, obs, A, wants, fold=False, live=None, zero_rest=True):
    """Several nets of one shape on the same inputs in ONE launch (rnad_mlp_forward_multi).  packed_list: their weight images;
    wants: per net (want_logits, want_value).  Returns a list of (logits [N, A] or None, value [N, 1] or None).
    live (FOLD kernels only): a row list -- only those rows are evaluated (the others: zeros, or uninitialised with zero_rest=False)."""
    fold = _fold_checked(fold, obs, "mlp_forward_multi")
    n = len(packed_list)
    assert 1 <= n <= 4 and len(wants) == n
    N = obs.numel() // (2 * A * A)
    half = obs.dtype == F16
    assert live is None or (fold and live.N == N), "a row list goes with the FOLD kernels"
    alloc = torch.zeros if (live is not None and zero_rest) else torch.empty
    outs = [(alloc((N, A), dtype=F32, device=obs.device) if wl else None,
             alloc((N, 1), dtype=F32, device=obs.device) if wv else None) for wl, wv in wants]
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    P = (C.c_void_p * n)(*[_dp(p, F32, "packed").value for p in packed_list])
    L = (C.c_void_p * n)(*[ptr(o[0]) for o in outs])
    V = (C.c_void_p * n)(*[ptr(o[1]) for o in outs])
    if fold:
        _check(lib().rnad_mlp_forward_fold(n, C.c_int64(N), *_row_list(live), A, W, P, _dp(obs, F16 if half else F32, "obs"), int(half), L, V,
                                           _stream()))
    else:
        _check(lib().rnad_mlp_forward_multi(n, C.c_int64(N), A, W, P, _dp(obs, F16 if half else F32, "obs"), int(half), L, V, _stream()))
    return outs


def mlp_backward_supported(A, W):
    return W % 32 == 0 and lib().rnad_mlp_backward_workspace(C.c_int64(32), A, W) > 0


def mlp_backward(packed, weights, obs, A, dlogits, dvalue, live=None, out=None, fold=False, capacity=None):
    """Gradients of the 8 Linear tensors (MLP_KEYS order) for dL/dlogits [N, A], dL/dvalue [N(,1)].
    live: a LiveRows -- only those rows contribute (the caller guarantees the others carry zero gradients).
    out: eight preallocated tensors shaped like the weights (e.g. views of one flat all-reduce bucket) to write into.
    fold: False, or the TreeHandle whose observation table `obs` is (as mlp_forward)."""
    fold = _fold_checked(fold, obs, "mlp_backward")
    N = obs.numel() // (2 * A * A)
    W = weights[0].shape[0]
    half = obs.dtype == F16
    grads = [torch.empty_like(w) for w in weights] if out is None else list(out)
    assert len(grads) == len(weights) and all(g.shape == w.shape for g, w in zip(grads, weights))
    ws = torch.empty((lib().rnad_mlp_backward_workspace(C.c_int64(N), A, W) // 4,), dtype=F32, device=obs.device)
    common = (A, W, _dp(packed, F32, "packed"), _dp(obs, F16 if half else F32, "obs"), int(half), _dp(dlogits, F32, "dlogits"),
              _dp(dvalue, F32, "dvalue"), *[_dp(g, F32, "grad") for g in grads], _dp(ws, F32, "workspace"), _stream())
    if fold:  # (packed: the fold image; gradients of the eight original tensors)
        assert live is None or live.N == N, "live-row list built for a different batch"
        # capacity: the length of a FIXED row list (the representatives of TreeHandle.obs_dedup): the launch -- workgroups, partial rows of
        # the reduction -- is then sized for it instead of for all N rows
        cap = N if (capacity is None or live is None) else int(capacity)
        _check(lib().rnad_mlp_backward_fold(C.c_int64(cap), *_row_list(live), *common))
    elif live is None:
        _check(lib().rnad_mlp_backward(C.c_int64(N), *common))
    else:
        assert live.N == N, "live-row list built for a different batch"
        _check(lib().rnad_mlp_backward_rows(C.c_int64(N), _dp(live.rows, I32, "rows"), C.c_void_p(live.count.data_ptr()), *common))
    return grads


class FusedMLP(torch.autograd.Function):
    """logits, value = FusedMLP.apply(obs, A, packed, *weights): rnad_mlp_forward / rnad_mlp_backward as one autograd node
    (`packed` = mlp_pack(weights, A); the weights themselves are only needed to route the gradients)."""

    @staticmethod
    def forward(ctx, obs, A, packed, *weights):
        ctx.A = A
        ctx.save_for_backward(obs, packed, *weights)
        return mlp_forward(packed, weights[0].shape[0], obs, A)

    @staticmethod
    def backward(ctx, dlogits, dvalue):
        obs, packed, *weights = ctx.saved_tensors
        grads = mlp_backward(packed, weights, obs, ctx.A, dlogits.contiguous(), dvalue.contiguous())
        return (None, None, None, *grads)


class FusedMLPRows(torch.autograd.Function):
    """FusedMLP restricted to a LiveRows list: logits, value = FusedMLPRows.apply(obs, A, packed, live, *weights).  Rows that
    are not listed come back as zeros and must receive zero gradients (they are not read in backward)."""

    @staticmethod
    def forward(ctx, obs, A, packed, live, *weights):
        ctx.A, ctx.live = A, live
        ctx.save_for_backward(obs, packed, *weights)
        return mlp_forward(packed, weights[0].shape[0], obs, A, live=live)

    @staticmethod
    def backward(ctx, dlogits, dvalue):
        obs, packed, *weights = ctx.saved_tensors
        grads = mlp_backward(packed, weights, obs, ctx.A, dlogits.contiguous(), dvalue.contiguous(), live=ctx.live)
        return (None, None, None, None, *grads)


# --------------------------------------------------------------------------------------- rollout driver
class Trajectory:
    """Preallocated [T_cap, B, ...] device buffers of one batch of episodes (struct rnad_traj)."""

    def __init__(self, tree, B, T_cap, device, half=False, with_observations=True, with_values=True, compact=False):
        """with_observations / with_values = False: those buffers are not allocated (bucketed rollout: observations are a function
        of (t & 1, indices) and are materialised on demand; the actor's values are only stored when asked for).
        compact=True (rollout_bucketed_compact): only `states` (the relative states below the cut, uint8 / int16 [T_cap + 1, B]: include/rnad_hip.h
        "Compact trajectory"), alive, `acts` (int64 [B], 3 bits per step) and `final_reward` [B] exist; `indices` (int32 [T_cap + 1, B], the
        reference's) is rebuilt by bucket_indices on first access, mask_bits / policy / actions / rewards are None until bucket_expand
        fills them."""
        A = tree.A
        self.T_cap, self.B, self.A, self.half = T_cap, B, A, half
        self.compact = bool(compact)
        self.device = torch.device(device)
        self._indices = None
        self._owner = None  # compact: (tree handle, Buckets) of the rollout that filled `states`
        if compact:
            plan = bucket_plan(tree, B)
            if plan is None:
                raise RnadHipError(lib().rnad_last_error().decode())
            self.states = torch.empty((T_cap + 1, B), dtype=U8 if plan.rel_bytes == 1 else torch.int16, device=device)
            self.observations = self.mask_bits = self.policy = self.actions = self.rewards = self.values = None
            self.acts = torch.empty((B,), dtype=torch.int64, device=device)
            self.final_reward = torch.empty((B,), dtype=F32, device=device)
            self.alive = torch.empty((T_cap + 1,), dtype=I32, device=device)
            self.c = None
            return
        self._indices = torch.empty((T_cap + 1, B), dtype=I32, device=device)
        self.observations = torch.empty((T_cap, B, 2, A, A), dtype=F16 if half else F32, device=device) if with_observations else None
        self.mask_bits = torch.empty((T_cap, B), dtype=U8, device=device)
        self.policy = torch.empty((T_cap, B, A), dtype=F32, device=device)
        self.actions = torch.empty((T_cap, B), dtype=I32, device=device)
        self.rewards = torch.empty((T_cap, B), dtype=F32, device=device)
        self.values = torch.empty((T_cap, B), dtype=F32, device=device) if with_values else None
        self.alive = torch.empty((T_cap + 1,), dtype=I32, device=device)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        self.c = Traj(T_cap, int(half), B, self._indices.data_ptr(), ptr(self.observations), self.mask_bits.data_ptr(),
                      self.policy.data_ptr(), self.actions.data_ptr(), self.rewards.data_ptr(), ptr(self.values),
                      self.alive.data_ptr())

    @property
    def indices(self):
        """int32 [T_cap + 1, B] (episode.py:218).  A compact trajectory rebuilds it from its relative states on first access
        (rnad_bucket_indices) and keeps it until invalidate() says the buffers were rewritten."""
        if self._indices is None and self.compact:
            if self._owner is None:
                raise RnadHipError("this compact trajectory has not been played yet")
            self._indices = bucket_indices(self._owner[0], self._owner[1], self)
        return self._indices

    def invalidate(self):
        """The rollout buffers were rewritten in place (a replayed hipGraph of the step): forget what was derived from them."""
        if self.compact:
            self._indices = None
            self.mask_bits = self.policy = self.actions = self.rewards = None


def rollout_begin(tree, traj):
    _check(lib().rnad_rollout_begin(tree.ptr, C.byref(traj.c), _stream()))


def rollout_run(tree, traj, W, packed, seed=0, lane0=0, keep_logits=False, skip_absorbed=False, store_values=True):
    """All T_cap steps of a rollout with the fused MLP as the actor, enqueued by one native call.
    keep_logits: return the actor's raw logits of every step as a [T_cap, B, A] tensor (else None).
    skip_absorbed: from step 1 on evaluate the actor only on lanes still in the tree (not with keep_logits).
    store_values=False: the actor's value head is not evaluated; traj.values is zeros."""
    dev = traj.indices.device
    logits = torch.zeros((traj.T_cap if keep_logits else 1, traj.B, tree.A), dtype=F32, device=dev)
    value = torch.zeros((traj.B,), dtype=F32, device=dev) if store_values else None
    null = C.c_void_p()
    rows = count = scratch = None
    if skip_absorbed:
        assert not keep_logits, "keep_logits needs the dense actor"
        rows = torch.empty((traj.B,), dtype=I32, device=dev)
        count = torch.empty((1,), dtype=torch.int64, device=dev)
        scratch = torch.empty((int(lib().rnad_compact_workspace(C.c_int64(traj.B))),), dtype=I32, device=dev)
    _check(lib().rnad_rollout_run(tree.ptr, C.byref(traj.c), int(W), _dp(packed, F32, "packed"), _dp(logits, F32, "logits"),
                                  C.c_int64(traj.B * tree.A if keep_logits else 0), _dp(value, F32, "value", True), C.c_uint64(seed),
                                  C.c_int64(lane0), _dp(rows, I32, "live_rows", True),
                                  C.c_void_p(count.data_ptr()) if count is not None else null,
                                  _dp(scratch, I32, "block_counts", True), _stream()))
    return logits if keep_logits else None


def rollout_run_tabular(tree, traj, logits_table, value_table=None, seed=0, lane0=0):
    """Rollout whose actor was evaluated once per (player, state): logits_table [2S, A], value_table [2S, 1] or None
    (rnad_rollout_run_tabular)."""
    assert logits_table.shape == (2 * tree.S, tree.A)
    _check(lib().rnad_rollout_run_tabular(tree.ptr, C.byref(traj.c), _dp(logits_table, F32, "logits_table"),
                                          _dp(value_table, F32, "value_table", True), C.c_uint64(seed), C.c_int64(lane0), _stream()))


def rollout_end(tree, traj):
    _check(lib().rnad_rollout_end(tree.ptr, C.byref(traj.c), _stream()))


def rollout_step(tree, traj, t, value, logits=None, policy=None, actions=None, noise_action=None, noise_chance=None,
                 seed=0, lane0=0):
    mode = 0 if logits is not None else (1 if actions is None else 2)
    _check(lib().rnad_rollout_step(tree.ptr, C.byref(traj.c), int(t), mode, _dp(logits, F32, "logits", True),
                                   _dp(policy, F32, "policy", True), _dp(actions, I32, "actions", True),
                                   _dp(value, F32, "value"), _dp(noise_action, F32, "noise_action", True),
                                   _dp(noise_chance, F32, "noise_chance", True), C.c_uint64(seed), C.c_int64(lane0), _stream()))


# --------------------------------------------------------------------------------------- learner kernels
def process_policy(policy, mask, n_disc, eps):
    A = policy.shape[-1]
    out = torch.empty_like(policy)
    _check(lib().rnad_process_policy(C.c_int64(policy.numel() // A), A, _dp(policy, F32, "policy"), _dp(mask, F32, "mask"),
                                     int(n_disc), C.c_float(eps), _dp(out, F32, "out"), _stream()))
    return out


def vtrace(v, valid, player_id, mu, pi, logpi, actions, reward, player, eta, lambda_, c, rho, gamma, want_has_played=True):
    """actions: one-hot f32 [T,B,A] or int32 [T,B]; player_id: int32 [T,B] or None (= t & 1)."""
    T, B, A = mu.shape
    onehot = actions.dtype == F32
    v_target = torch.empty((T, B), dtype=F32, device=mu.device)
    q = torch.empty((T, B, A), dtype=F32, device=mu.device)
    has_played = torch.empty((T, B), dtype=I32, device=mu.device) if want_has_played else None
    _check(lib().rnad_vtrace(T, C.c_int64(B), A, _dp(v, F32, "v"), _dp(valid, F32, "valid"), _dp(player_id, I32, "player_id", True),
                             _dp(mu, F32, "mu"), _dp(pi, F32, "pi"), _dp(logpi, F32, "logpi"),
                             _dp(actions, F32 if onehot else I32, "actions"), int(onehot), _dp(reward, F32, "reward"), int(player),
                             C.c_float(eta), C.c_float(lambda_), C.c_float(c), C.c_float(rho), C.c_float(gamma),
                             _dp(v_target, F32, "v_target"), _dp(has_played, I32, "has_played", True), _dp(q, F32, "q"), _stream()))
    return v_target, has_played, q


def mask_sum(mask):
    out = torch.empty((1,), dtype=F64, device=mask.device)
    _check(lib().rnad_mask_sum(C.c_int64(mask.numel()), _dp(mask, F32, "mask"), _dp(out, F64, "out"), _stream()))
    return out


def loss_v(v, v_target, mask, norm, weight, loss, dv, accumulate):
    _check(lib().rnad_loss_v(C.c_int64(v.numel()), _dp(v, F32, "v"), _dp(v_target, F32, "v_target"), _dp(mask, F32, "mask"),
                             _dp(norm, F64, "norm"), C.c_float(weight), _dp(loss, F64, "loss", True), _dp(dv, F32, "dv", True),
                             int(accumulate), _stream()))


def loss_nerd(logit, pi, q, mask, legal, norm, clip, threshold, weight, loss, dlogit, accumulate):
    A = logit.shape[-1]
    _check(lib().rnad_loss_nerd(C.c_int64(logit.numel() // A), A, _dp(logit, F32, "logit"), _dp(pi, F32, "pi"), _dp(q, F32, "q"),
                                _dp(mask, F32, "mask"), _dp(legal, F32, "legal"), _dp(norm, F64, "norm"), C.c_float(clip),
                                C.c_float(threshold), C.c_float(weight), _dp(loss, F64, "loss", True),
                                _dp(dlogit, F32, "dlogit", True), int(accumulate), _stream()))


def learn_fused(indices, mask_bits, actions, rewards, mu, logit, v, v_target_net, logit_reg, logit_reg_, norm, hp,
                want_aux=False):
    """One pass over a [T,B] trajectory: returns dlogit [T,B,A], dv [T,B], losses f64[2] and (optionally) pi, v_target, q."""
    T, B, A = mu.shape
    dev = mu.device
    dlogit = torch.empty((T, B, A), dtype=F32, device=dev)
    dv = torch.empty((T, B), dtype=F32, device=dev)
    losses = torch.empty((2,), dtype=F64, device=dev)
    pi = torch.empty((T, B, A), dtype=F32, device=dev) if want_aux else None
    vt = torch.empty((2, T, B), dtype=F32, device=dev) if want_aux else None
    q = torch.empty((2, T, B, A), dtype=F32, device=dev) if want_aux else None
    _check(lib().rnad_learn_fused(T, C.c_int64(B), A, _dp(indices, I32, "indices"), _dp(mask_bits, U8, "mask_bits"),
                                  _dp(actions, I32, "actions"), _dp(rewards, F32, "rewards"), _dp(mu, F32, "mu"),
                                  _dp(logit, F32, "logit"), _dp(v, F32, "v"), _dp(v_target_net, F32, "v_target_net"),
                                  _dp(logit_reg, F32, "logit_reg"), _dp(logit_reg_, F32, "logit_reg_"), _dp(norm, F64, "norm"),
                                  C.byref(hp), _dp(losses, F64, "losses"), _dp(dlogit, F32, "dlogit"), _dp(dv, F32, "dv"),
                                  _dp(pi, F32, "pi", True), _dp(vt, F32, "vt", True), _dp(q, F32, "q", True), _stream()))
    return dlogit, dv, losses, pi, vt, q


def learn_fused_tabular(tree, indices, mask_bits, actions, rewards, mu, logit_tab, v_tab, v_target_tab, logit_reg_tab, logit_reg_tab_,
                        norm, hp):
    """rnad_learn_fused_tabular: net outputs given per (player, state) row [2S, (A)]; returns dlogit_tab [2S, A], dv_tab [2S, 1]
    (the per-slot gradients summed per row) and losses f64[2]."""
    T, B, A = mu.shape
    dev = mu.device
    S = tree.S
    ws = torch.empty((int(lib().rnad_learn_tabular_workspace(tree.ptr, T, C.c_int64(B))) // 8 + 1,), dtype=F64, device=dev)
    dlogit = torch.empty((2 * S, A), dtype=F32, device=dev)
    dv = torch.empty((2 * S, 1), dtype=F32, device=dev)
    losses = torch.empty((2,), dtype=F64, device=dev)
    _check(lib().rnad_learn_fused_tabular(tree.ptr, T, C.c_int64(B), _dp(indices, I32, "indices"), _dp(mask_bits, U8, "mask_bits"),
                                          _dp(actions, I32, "actions"), _dp(rewards, F32, "rewards"), _dp(mu, F32, "mu"),
                                          _dp(logit_tab, F32, "logit_tab"), _dp(v_tab, F32, "v_tab"), _dp(v_target_tab, F32, "v_target_tab"),
                                          _dp(logit_reg_tab, F32, "logit_reg_tab"), _dp(logit_reg_tab_, F32, "logit_reg_tab_"),
                                          _dp(norm, F64, "norm"), C.byref(hp), _dp(losses, F64, "losses"), _dp(ws, F64, "workspace"),
                                          _dp(dlogit, F32, "dlogit_tab"), _dp(dv, F32, "dv_tab"), _stream()))
    return dlogit, dv, losses


def row_sums(tree, indices, dlogit, dv):
    """Per-(player, state)-row sums of per-slot gradients dlogit [T,B,A], dv [T,B] (rnad_row_sums) -> [2S, A], [2S, 1]."""
    T, B = indices.shape
    A, S, dev = tree.A, tree.S, indices.device
    ws = torch.empty((int(lib().rnad_row_sums_workspace(tree.ptr)) // 8 + 1,), dtype=F64, device=dev)
    out_l = torch.empty((2 * S, A), dtype=F32, device=dev)
    out_v = torch.empty((2 * S, 1), dtype=F32, device=dev)
    _check(lib().rnad_row_sums(tree.ptr, T, C.c_int64(B), _dp(indices, I32, "indices"), _dp(dlogit, F32, "dlogit"), _dp(dv, F32, "dv"),
                               _dp(ws, F64, "workspace"), _dp(out_l, F32, "dlogit_tab"), _dp(out_v, F32, "dv_tab"), _stream()))
    return out_l, out_v


class TabularMLP(torch.autograd.Function):
    """logits [T*B, A], value [T*B, 1] of an MLP on a trajectory whose observations are rows of the tree's observation table:
    the net is evaluated on the 2S rows and every slot gathers its row; backward sums the per-slot gradients per row
    (rnad_row_sums) and differentiates through the 2S evaluations.  apply(table, indices [T,B] int32, tree, A, packed, *weights)."""

    @staticmethod
    def forward(ctx, table, indices, tree, A, packed, *weights):
        T, B = indices.shape
        logit_tab, v_tab = mlp_forward(packed, weights[0].shape[0], table, A)
        rows = (indices.long() + (torch.arange(T, device=indices.device) & 1).view(T, 1) * tree.S).reshape(-1)
        ctx.A, ctx.tree = A, tree
        ctx.save_for_backward(table, indices, packed, *weights)
        return logit_tab.index_select(0, rows), v_tab.index_select(0, rows)

    @staticmethod
    def backward(ctx, dlogits, dvalue):
        table, indices, packed, *weights = ctx.saved_tensors
        T, B = indices.shape
        dl_tab, dv_tab = row_sums(ctx.tree, indices, dlogits.contiguous().view(T, B, ctx.A), dvalue.contiguous().view(T, B))
        grads = mlp_backward(packed, weights, table, ctx.A, dl_tab, dv_tab)
        return (None, None, None, None, None, *grads)


def learn_fused_gather(tree, indices, mask_bits, actions, rewards, mu, logit_tab, v_tab, v_target_tab, logit_reg_tab, logit_reg_tab_,
                       norm, hp):
    """rnad_learn_fused_gather: net outputs given per (player, state) row; returns per-slot dlogit [T,B,A], dv [T,B], losses."""
    T, B, A = mu.shape
    dev = mu.device
    dlogit = torch.empty((T, B, A), dtype=F32, device=dev)
    dv = torch.empty((T, B), dtype=F32, device=dev)
    losses = torch.empty((2,), dtype=F64, device=dev)
    ws = torch.empty((int(lib().rnad_learn_gather_workspace(tree.ptr)) // 4,), dtype=F32, device=dev)
    _check(lib().rnad_learn_fused_gather(tree.ptr, T, C.c_int64(B), _dp(indices, I32, "indices"), _dp(mask_bits, U8, "mask_bits"),
                                         _dp(actions, I32, "actions"), _dp(rewards, F32, "rewards"), _dp(mu, F32, "mu"),
                                         _dp(logit_tab, F32, "logit_tab"), _dp(v_tab, F32, "v_tab"), _dp(v_target_tab, F32, "v_target_tab"),
                                         _dp(logit_reg_tab, F32, "logit_reg_tab"), _dp(logit_reg_tab_, F32, "logit_reg_tab_"),
                                         _dp(norm, F64, "norm"), C.byref(hp), _dp(losses, F64, "losses"), _dp(ws, F32, "workspace"),
                                         _dp(dlogit, F32, "dlogit"), _dp(dv, F32, "dv"), _stream()))
    return dlogit, dv, losses


# --------------------------------------------------------------------------------------- bucketed tabular pipeline
class BucketPlan:
    """rnad_bucket_plan for (tree, B): partition depth, list capacities and workspace sizes; plus the device workspaces themselves
    (`scratch` of the rollout, zero-initialised `accumulators` of the learner), allocated once per (tree, B) and reused."""

    def __init__(self, tree, B, out):
        self.B = B
        (self.rows, self.n_buckets, self.n_upper, self.n_groups, self.max_items, self.scratch_bytes, self.acc_bytes, self.lds,
         self.rel_bytes, self.sort_tile, self.chunk) = (int(x) for x in out)
        dev = tree.device
        self.scratch = torch.empty((self.scratch_bytes // 4 + 1,), dtype=I32, device=dev)
        self.accumulators = torch.zeros((self.acc_bytes // 8 + 1,), dtype=torch.int64, device=dev)


_owner = threading.local()


class WorkspaceToken:
    """Identity of one owner of bucketed-pipeline workspaces (see workspace_owner); the workspaces live as long as the token."""

    __slots__ = ("__weakref__",)


class workspace_owner:
    """`with workspace_owner(token):` -- the BucketPlans made or looked up inside belong to `token` (a WorkspaceToken; every RNaD object
    holds one): their device workspaces (`scratch` of the sort and the rollout, the learner's `accumulators`, the staging buffers) are that
    owner's alone.  Calls outside any such block share the default set, as every call did before r05 -- fine for one trainer per
    (tree, batch size), and for several whose calls are ordered by one stream; two trainers stepping on two streams of one device
    (reference main.py:55-81 runs several over one tree) each need their own, or they add into each other's sums."""

    def __init__(self, token):
        self.token, self.prev = token, None

    def __enter__(self):
        self.prev = getattr(_owner, "token", None)
        _owner.token = self.token
        return self

    def __exit__(self, *exc):
        _owner.token = self.prev
        return False


def plan_knobs():
    """The tuning overrides csrc/bucket.hip's make_plan reads from the environment on every call: they size the plan's workspaces, so they
    key every cache of plans (and RNaD's captured graphs)."""
    return tuple(os.environ.get(k) for k in ("RNAD_BUCKET_ROWS", "RNAD_BUCKET_CHUNK", "RNAD_SORT_TILE"))


def bucket_plan(tree, B):
    """The BucketPlan of (tree, B) for the current workspace_owner (cached on the tree handle), or None when this tree / batch cannot be
    bucketed."""
    destroy_deferred()
    token = getattr(_owner, "token", None)
    if token is None:
        cache = tree.__dict__.setdefault("_bucket_plans", {})
    else:  # (weakly keyed: a trainer's workspaces go when the trainer does)
        cache = tree.__dict__.setdefault("_owner_plans", weakref.WeakKeyDictionary()).setdefault(token, {})
    key = (B,) + plan_knobs()
    if key not in cache:
        out = (C.c_int64 * 11)()
        rc = lib().rnad_bucket_plan(tree.ptr, B, out)
        cache[key] = BucketPlan(tree, B, list(out)) if rc == 0 else None
    return cache[key]


def bucket_map(tree, B):
    """(bucket_of int32 [S] CPU tensor, n_groups): the bucket of every state under the plan of (tree, B) (rnad_bucket_map)."""
    out = torch.empty((tree.S,), dtype=I32)
    n_groups = C.c_int32()
    _check(lib().rnad_bucket_map(tree.ptr, B, C.c_void_p(out.data_ptr()), C.byref(n_groups)))
    return out, n_groups.value


def bucket_shared_steps(tree, B):
    """int32 [n_buckets] CPU tensor: the env steps a lane of each bucket shares with its whole bucket (rnad_bucket_shared_steps)."""
    plan = bucket_plan(tree, B)
    out = torch.empty((plan.n_buckets,), dtype=I32)
    _check(lib().rnad_bucket_shared_steps(tree.ptr, B, C.c_void_p(out.data_ptr())))
    return out


def stored_state_slots(tree, buckets, T):
    """Slots of a compact trajectory that hold a relative state: per column the rows n_shared(bucket) .. T (what the rollout writes and
    the learner reads; bench.py's algorithmic bytes)."""
    n = int(buckets.n_items.item())
    items = buckets.items[:n].cpu().long()
    shared = bucket_shared_steps(tree, buckets.plan.B).long()[items[:, 2]]
    return int((items[:, 1] * (T + 1 - shared).clamp(min=0)).sum().item())


class Buckets:
    """Lane permutation and learner work list of one bucket-ordered batch (outputs of rnad_rollout_bucketed)."""

    def __init__(self, plan, device):
        self.plan = plan
        self.lane_ids = torch.empty((plan.B,), dtype=I32, device=device)
        self.items = torch.empty((plan.max_items, 4), dtype=I32, device=device)
        self.n_items = torch.empty((1,), dtype=I32, device=device)
        self.norm = torch.empty((2,), dtype=F64, device=device)  # N_P of the batch: live slots of parity P
        self.alive_pending = None  # the compact Trajectory whose alive counts / norm are still un-summed (rollout_bucketed_compact(defer_alive=True))


def rollout_bucketed(tree, traj, table, value_table=None, seed=0, lane0=0, table_is_policy=False, column=0, step_params=None):
    """rnad_rollout_bucketed.  table [2S, stride]: the tabular actor per (player, state) row, A floats starting at `column` -- its
    logits (table_is_policy=False: a [2S, A] logits table) or its policy (True: e.g. the pi columns of bucket_records, see
    policy_column).  value_table [2S, 1] or None.  Returns the Buckets of the batch."""
    plan = bucket_plan(tree, traj.B)
    if plan is None:
        raise RnadHipError(lib().rnad_last_error().decode())
    assert table.shape[0] == 2 * tree.S and table.shape[1] >= column + tree.A
    buckets = Buckets(plan, traj.device)
    base = _dp(table, F32, "table")
    _check(lib().rnad_rollout_bucketed(tree.ptr, C.byref(traj.c), C.c_void_p(base.value + 4 * column), table.shape[1], int(table_is_policy),
                                       _dp(value_table, F32, "value_table", True), 1, seed, lane0,
                                       _dp(step_params, torch.int64, "step_params", True), _dp(plan.scratch, I32, "scratch"),
                                       _dp(buckets.lane_ids, I32, "lane_ids"), _dp(buckets.items, I32, "items"),
                                       _dp(buckets.n_items, I32, "n_items"), _dp(buckets.norm, F64, "norm"), _stream()))
    return buckets


COMPACT_MAX_STEPS = 21  # 3 bits of action per step in one 64-bit word (csrc/bucket.hip kCompactSteps)
BUCKET_REPLICAS = 64  # copies of the upper rows' accumulators (csrc/bucket.hip kReplicas; rnad_bucket_plan out[6] counts them)
BUCKET_MAX_LANES = 1 << 22  # lanes per call of the bucketed pipeline (csrc/bucket.hip kLaneBits: fixed-point headroom of the row sums)


def rollout_bucketed_compact(tree, traj, table, seed=0, lane0=0, step_params=None, table_is_policy=True, column=None, visited=None,
                             defer_alive=False):
    """rnad_rollout_bucketed_compact: the episodes of rollout_bucketed(table, table_is_policy, column) into a Trajectory(compact=True).
    table: bucket_records(...) (the default: its pi columns are the actor), any [2S, stride] table with the actor's policy rows from
    `column` on, or (table_is_policy=False) the actor's logits [2S, A].  visited (int32 [2S], optional): set to 1 for every (player,
    state) row a live slot sits in (and the two rows of the absorbing state), 0 elsewhere.  defer_alive: `traj.alive` and the
    normalisers `buckets.norm` are left to the learner (learn_bucketed_compact adds them up in its own launch: one kernel less per
    step) or to bucket_alive(); `buckets.alive_pending` says so until one of them ran.  Returns the Buckets of the batch."""
    assert traj.compact and traj.T_cap <= COMPACT_MAX_STEPS
    plan = bucket_plan(tree, traj.B)
    if plan is None:
        raise RnadHipError(lib().rnad_last_error().decode())
    given = table  # (a records tensor may carry a pending rows_expand job: see below)
    pol_rows = getattr(table, "_policy_rows", None)
    if table_is_policy and column is None and pol_rows is not None:
        table, column = pol_rows, 0  # the same floats as the pi columns of the records, 16 bytes per row
    if column is None:
        column = policy_column(tree.A) if table_is_policy else 0
    assert table.shape[0] == 2 * tree.S and table.shape[1] >= column + tree.A
    assert visited is None or visited.numel() == 2 * tree.S
    buckets = Buckets(plan, traj.device)
    _complete_pending(tree, plan)
    base = _dp(table, F32, "table")
    pending = getattr(given, "_expand", None) if table_is_policy else None
    if pending is not None:
        # distinct observations: the tables hold their representatives' rows only -- the copies to the other rows ride in the keys pass
        dedup, tabs = pending
        tabs = [t for t in tabs if t is not None]
        ptrs = (C.c_void_p * len(tabs))(*[_dp(t, F32, "table").value for t in tabs])
        widths = (C.c_int32 * len(tabs))(*[int(t.shape[1]) for t in tabs])
        _check(lib().rnad_rollout_bucketed_compact_expand(
            tree.ptr, traj.T_cap, traj.B, C.c_void_p(base.value + 4 * column), table.shape[1], int(table_is_policy), seed, lane0,
            _dp(step_params, torch.int64, "step_params", True), _dp(plan.scratch, I32, "scratch"), _dp(buckets.lane_ids, I32, "lane_ids"),
            _dp(buckets.items, I32, "items"), _dp(buckets.n_items, I32, "n_items"), _dp(buckets.norm, F64, "norm"),
            _dp(traj.states, traj.states.dtype, "states"), None if defer_alive else _dp(traj.alive, I32, "alive"),
            _dp(traj.acts, torch.int64, "acts"), _dp(traj.final_reward, F32, "final_reward"), _dp(visited, I32, "visited", True),
            _dp(dedup.rep_of, I32, "rep_of"), len(tabs), ptrs, widths, _stream()))
        given._expand = None
    else:
        _rollout_compact_plain(tree, traj, table, base, column, table_is_policy, seed, lane0, step_params, plan, buckets, defer_alive, visited)
    buckets.alive_pending = traj if defer_alive else None
    plan._pending = buckets if defer_alive else None
    traj._owner = (tree, buckets)
    traj.invalidate()
    return buckets


PLAY_LEARN_FINISH, PLAY_LEARN_DISTINCT = 1, 2  # include/rnad_hip.h


class _LeafPathsC(C.Structure):
    """struct rnad_leaf_paths (include/rnad_hip.h)."""

    _fields_ = [("n_cols", C.c_int64), ("rows", C.c_int32), ("T_cap", C.c_int32), ("states", C.c_void_p), ("acts", C.c_void_p),
                ("final_reward", C.c_void_p), ("items", C.c_void_p), ("n_items", C.c_void_p), ("max_items", C.c_int32),
                ("col_of", C.c_void_p), ("col_count", C.c_void_p), ("bucket_col0", C.c_void_p), ("crowded_lanes", C.c_int32)]


LEAF_CROWDED_SHARE = 4.0  # a bucket with this many times its even share of the batch's lanes is counted by the rollout's work items (r06)


class LeafPaths:
    """The tree's leaf paths as a batch in the compact layout (struct rnad_leaf_paths): one column per terminal transition (state, row
    action, column action, outcome) of a reachable state, sorted by the bucket of the cut that batches of `plan_B` lanes get -- the
    trajectory every lane that leaves the tree by that transition has played (tree.py:311-330: a state has one parent entry) --, its work
    items (<= 256 columns of one bucket) and `col_of` (transition -> column).  Built once per (tree, cut) from the tree's tensors in the reference layout (index int64 [S, C, A, A], chance, value f32 [S, C, A, A]); see leaf_paths()."""

    def __init__(self, tree, plan_B, index, chance, value):
        dev = tree.device
        S, Cc, A = tree.S, tree.C, tree.A
        T_cap = 2 * tree.max_depth
        assert T_cap <= COMPACT_MAX_STEPS
        bucket_of, n_groups = bucket_map(tree, plan_B)
        cols = leaf_columns(index.to(dev), chance.to(dev), value.to(dev), bucket_of.to(dev), tree.max_depth)
        n = cols["n_cols"]
        self.n_cols, self.T_cap, self.plan_B = n, T_cap, plan_B
        indices = cols["indices"]
        self.acts, self.final_reward, self.col_of = cols["acts"], cols["final_reward"], cols["col_of"]
        # work items: <= 256 consecutive columns of one bucket
        items = leaf_items(cols["bucket"], max(64, min(256, int(os.environ.get("RNAD_LEAF_CHUNK", "256")))))  # (tuning knob: columns per work item)
        self.max_items = len(items)
        self.items = torch.tensor(items, dtype=I32, device=dev).contiguous()
        self.n_items = torch.tensor([len(items)], dtype=I32, device=dev)
        # relative states under that cut (1 or 2 bytes per slot)
        bad = torch.zeros((1,), dtype=I32, device=dev)
        rows, rel = C.c_int32(), C.c_int32()
        states8 = torch.empty((T_cap + 1, n), dtype=torch.int16, device=dev)  # (room for the two-byte case)
        _check(lib().rnad_leaf_paths_pack(tree.ptr, C.c_int64(plan_B), T_cap + 1, C.c_int64(n), _dp(indices, I32, "indices"),
                                          _dp(self.items, I32, "items"), _dp(self.n_items, I32, "n_items"), self.max_items,
                                          C.c_void_p(states8.data_ptr()), _dp(bad, I32, "mismatch"), C.byref(rows), C.byref(rel), _stream()))
        if int(bad.item()) != 0:
            raise RnadHipError("leaf paths: a column does not lie in its work item's bucket")
        self.rows, self.rel_bytes = rows.value, rel.value
        self.states = states8
        self.indices = indices  # (tests: the dense states of the columns)
        # r06: who counts a bucket's lanes per column (rnad_leaf_paths_t.col_count): the learner's work items, each scanning the bucket's lanes
        # -- or, for a bucket that holds LEAF_CROWDED_SHARE times its even share of the batch (a sharpened policy), the rollout's work items
        # with a histogram in LDS.  RNAD_LEAF_CROWDED_LANES forces the threshold (1: every bucket, 0: none -- tests).
        plan = bucket_plan(tree, plan_B)
        self.col_count = torch.zeros((n,), dtype=I32, device=dev)
        self.bucket_col0 = torch.searchsorted(cols["bucket"].long().contiguous(), torch.arange(plan.n_buckets + 1, device=dev)).to(I32).contiguous()
        forced = os.environ.get("RNAD_LEAF_CROWDED_LANES")
        self.crowded_lanes = int(forced) if forced is not None else max(2048, int(LEAF_CROWDED_SHARE * plan_B / max(n_groups, 1)))
        self.c = _LeafPathsC(n, self.rows, T_cap, states8.data_ptr(), self.acts.data_ptr(), self.final_reward.data_ptr(), self.items.data_ptr(),
                             self.n_items.data_ptr(), self.max_items, self.col_of.data_ptr(), self.col_count.data_ptr(),
                             self.bucket_col0.data_ptr(), self.crowded_lanes)


def leaf_columns(index, chance, value, bucket_of, max_depth):
    """The tree's terminal transitions as trajectories (pure tensor code: any device; tests/test_leaf_columns.py runs it on the CPU).  index int64 / chance / value f32 [S, C, A, A] in the reference layout (tree.py:115-146), bucket_of int [S]
    (< 0: unreachable).  One column per (state s >= 1, outcome c, row action a0, column action a1) with index == 0 and chance > 0, sorted
    by (bucket, state, code) with code = (a0 * A + a1) * C + c.  Returns n_cols, bucket [n] (the column's bucket), indices int32
    [2 * max_depth + 1, n] (the states of its env steps, 0 once the episode is over: episode.py:96-125), acts int64 [n] (3 bits per env
    step), final_reward f32 [n] (the transition's value: rewards *= (indices == 0), episode.py:120-121) and col_of int32 [S * A * A * C]
    (transition ((s * A + a0) * A + a1) * C + c -> column, -1 if it is not terminal)."""
    S, Cc, A, _ = index.shape
    dev = index.device
    index = index.long()
    live = chance > 0
    bucket_of = bucket_of.long()
    T_cap = 2 * max_depth
    # parent entry of every state but the root: the (only) transition that leads to it (ids are DFS pre-order, tree.py:311-330)
    s_, c_, a0_, a1_ = torch.nonzero((index != 0) & live, as_tuple=True)
    child = index[s_, c_, a0_, a1_]
    parent = torch.zeros((S,), dtype=torch.long, device=dev)
    pa0, pa1 = torch.zeros_like(parent), torch.zeros_like(parent)
    parent[child], pa0[child], pa1[child] = s_, a0_, a1_
    # terminal transitions of reachable states (state 0 is the absorbing state: not a state of the tree)
    ts, tc, ta0, ta1 = torch.nonzero((index == 0) & live, as_tuple=True)
    keep = (ts != 0) & (bucket_of[ts] >= 0)
    ts, tc, ta0, ta1 = ts[keep], tc[keep], ta0[keep], ta1[keep]
    code = (ta0 * A + ta1) * Cc + tc
    order = torch.argsort((bucket_of[ts] * S + ts) * (A * A * Cc) + code)  # by bucket, then state, then outcome (all distinct)
    ts, tc, ta0, ta1, code = ts[order], tc[order], ta0[order], ta1[order], code[order]
    n = int(ts.numel())
    assert n >= 1, "a tree without terminal transitions"
    # depth of the last state of every column, then its path up to the root
    depth = torch.zeros((n,), dtype=torch.long, device=dev)
    cur = ts.clone()
    for _ in range(max_depth):
        up = cur != 1
        depth += up.long()
        cur = torch.where(up, parent[cur], cur)
    assert bool((cur == 1).all()), "every reachable state descends from state 1"
    indices = torch.zeros((T_cap + 1, n), dtype=torch.int32, device=dev)
    acts = torch.zeros((n,), dtype=torch.int64, device=dev)
    cols = torch.arange(n, device=dev)
    cur, a0, a1 = ts.clone(), ta0.clone(), ta1.clone()
    for _ in range(max_depth):
        on = depth >= 0
        d = depth.clamp(min=0)
        st = torch.where(on, cur, torch.zeros_like(cur)).to(torch.int32)
        indices[2 * d[on], cols[on]] = st[on]
        indices[2 * d[on] + 1, cols[on]] = st[on]
        acts += torch.where(on, (a0 << (6 * d)) | (a1 << (6 * d + 3)), torch.zeros_like(a0))  # 3 bits per env step: row step 2d, column step 2d + 1
        a0, a1 = torch.where(on, pa0[cur], a0), torch.where(on, pa1[cur], a1)
        cur = torch.where(on, parent[cur], cur)
        depth = depth - 1
    col_of = torch.full((S * A * A * Cc,), -1, dtype=torch.int32, device=dev)
    col_of[ts * (A * A * Cc) + code] = cols.to(torch.int32)
    return dict(n_cols=n, bucket=bucket_of[ts], indices=indices, acts=acts.contiguous(),
                final_reward=value[ts, tc, ta0, ta1].to(torch.float32).contiguous(), col_of=col_of)


def leaf_items(bucket, chunk=256):
    """Work items over columns sorted by bucket: (begin, count, bucket, single) with at most `chunk` columns of ONE bucket each, a bucket's
    columns in equal shares (729 columns are 3 x 243, not 256 + 256 + 217)."""
    uniq, counts = torch.unique_consecutive(bucket, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    items = []
    for bk, st0, cnt in zip(uniq.tolist(), starts.tolist(), counts.tolist()):
        chunks = (cnt + chunk - 1) // chunk
        per = (cnt + chunks - 1) // chunks
        for k in range(chunks):
            items.append((st0 + per * k, min(per, cnt - per * k), bk, int(chunks == 1)))
    return items


def leaf_paths(tree, plan_B, index, chance, value):
    """The LeafPaths of (tree, the cut of batches of plan_B lanes), cached on the BucketPlan.  index / chance / value: the tree's tensors in
    the reference layout."""
    plan = bucket_plan(tree, plan_B)
    if plan is None:
        return None
    cache = plan.__dict__.setdefault("leaf_by_chunk", {})
    key = (os.environ.get("RNAD_LEAF_CHUNK"), os.environ.get("RNAD_LEAF_CROWDED_LANES"))  # (tuning knobs that are part of what a LeafPaths is)
    got = cache.get(key)
    if got is None:
        got = cache[key] = LeafPaths(tree, plan_B, index, chance, value)
    plan.leaf = got
    return got


def rollout_learn_bucketed_compact(tree, traj, records, fast_records, hp, seed=0, lane0=0, step_params=None, norm_is_global=True, rows=None,
                                   groups=None, distinct=False, norm_global=None, leaf=None):
    """rnad_rollout_learn_bucketed_compact: rollout_bucketed_compact(records) and learn_bucketed_compact of the batch it plays (T = T_cap)
    with ONE launch for rollout + learner -- the trajectory, traj.alive, buckets.norm and the per-row gradient tables of the two calls, bit
    for bit.  records: bucket_records(..., fast=True)[0] (its policy rows are the actor; a pending rows_expand job rides in the keys pass).
    norm_is_global=False (data parallel): stops before k_bucket_finish -- all-reduce buckets.norm, then bucket_finish(...).
    distinct: the learner half once per distinct trajectory of a (larger) work item, weighted with its lanes (RNAD_PLAY_LEARN_DISTINCT).
    norm_global (f64 [2], device): the finish divides by these normalisers -- those of a global batch -- instead of the batch's own
    (buckets.norm still receives the batch's own counts); implies the finish.
    leaf (LeafPaths of this tree and batch size): rollout and learner as two launches, the learner on the tree's leaf paths weighted with
    the lanes that played them -- the same per-row sums bit for bit, at a cost that does not grow with the batch.
    Returns (buckets, dlogit, dv); the tables are None when the finish is left to the caller."""
    assert traj.compact and traj.T_cap <= COMPACT_MAX_STEPS
    assert leaf is None or (leaf.plan_B == traj.B and leaf.T_cap == traj.T_cap and not distinct)
    plan = bucket_plan(tree, traj.B)
    if plan is None:
        raise RnadHipError(lib().rnad_last_error().decode())
    table = getattr(records, "_policy_rows", None)
    column = 0
    if table is None:
        table, column = records, policy_column(tree.A)
    assert table.shape[0] == 2 * tree.S and table.shape[1] >= column + tree.A
    buckets = Buckets(plan, traj.device)
    _complete_pending(tree, plan)
    rows, groups = _rows_and_groups(tree, buckets, rows, groups)
    base = _dp(table, F32, "table")
    pending = getattr(records, "_expand", None)
    n_tabs, ptrs, widths, rep_of = 0, None, None, None
    if pending is not None:
        dedup, tabs = pending
        tabs = [t for t in tabs if t is not None]
        n_tabs = len(tabs)
        ptrs = (C.c_void_p * n_tabs)(*[_dp(t, F32, "table").value for t in tabs])
        widths = (C.c_int32 * n_tabs)(*[int(t.shape[1]) for t in tabs])
        rep_of = _dp(dedup.rep_of, I32, "rep_of")
    dev, A = traj.device, tree.A
    norm_is_global = bool(norm_is_global) or norm_global is not None
    dlogit = torch.empty((2 * tree.S, A), dtype=F32, device=dev) if norm_is_global else None
    dv = torch.empty((2 * tree.S, 1), dtype=F32, device=dev) if norm_is_global else None
    _check(lib().rnad_rollout_learn_bucketed_compact(
        tree.ptr, traj.T_cap, traj.B, C.c_void_p(base.value + 4 * column), table.shape[1], seed, lane0,
        _dp(step_params, torch.int64, "step_params", True), _dp(plan.scratch, I32, "scratch"), _dp(buckets.lane_ids, I32, "lane_ids"),
        _dp(buckets.items, I32, "items"), _dp(buckets.n_items, I32, "n_items"), _dp(buckets.norm, F64, "norm"),
        _dp(traj.states, traj.states.dtype, "states"), _dp(traj.alive, I32, "alive"), _dp(traj.acts, torch.int64, "acts"),
        _dp(traj.final_reward, F32, "final_reward"), rep_of, n_tabs, ptrs, widths, _dp(fast_records, F32, "fast_records"), C.byref(hp),
        _dp(plan.accumulators, torch.int64, "accumulators"),
        (PLAY_LEARN_FINISH if norm_is_global else 0) | (PLAY_LEARN_DISTINCT if distinct else 0), _dp(norm_global, F64, "norm_global", True),
        _dp(dlogit, F32, "dlogit_tab", True),
        _dp(dv, F32, "dv_tab", True), *_row_list(rows), groups, C.byref(leaf.c) if leaf is not None else None, _stream()))
    if pending is not None:
        records._expand = None
    buckets.alive_pending = None
    plan._pending = None
    traj._owner = (tree, buckets)
    traj.invalidate()
    return buckets, dlogit, dv


def _rollout_compact_plain(tree, traj, table, base, column, table_is_policy, seed, lane0, step_params, plan, buckets, defer_alive, visited):
    _check(lib().rnad_rollout_bucketed_compact(tree.ptr, traj.T_cap, traj.B, C.c_void_p(base.value + 4 * column), table.shape[1],
                                               int(table_is_policy), seed, lane0, _dp(step_params, torch.int64, "step_params", True),
                                               _dp(plan.scratch, I32, "scratch"), _dp(buckets.lane_ids, I32, "lane_ids"),
                                               _dp(buckets.items, I32, "items"), _dp(buckets.n_items, I32, "n_items"),
                                               _dp(buckets.norm, F64, "norm"), _dp(traj.states, traj.states.dtype, "states"),
                                               None if defer_alive else _dp(traj.alive, I32, "alive"),
                                               _dp(traj.acts, torch.int64, "acts"), _dp(traj.final_reward, F32, "final_reward"),
                                               _dp(visited, I32, "visited", True), _stream()))


def _complete_pending(tree, plan, buckets=None):
    """The un-summed alive partials of a rollout with defer_alive=True live in plan.scratch, which every Buckets of the plan shares: before
    another rollout of the same (tree, B) overwrites them, the counts of the batch that still waits for them are added up (a launch; RNaD's
    own flow never gets here -- its learner completes the batch within the step)."""
    prev = getattr(plan, "_pending", None)
    if prev is not None and prev is not buckets and prev.alive_pending is not None:
        bucket_alive(tree, prev)
    plan._pending = None


def bucket_upper_rows(tree, B):
    """RowList of the (player, state) rows a staged actor must have evaluated BEFORE bucket_sort: both rows of every upper state of
    the cut of (tree, B) and of the absorbing state.  Cached on the plan."""
    plan = bucket_plan(tree, B)
    if getattr(plan, "upper_rows", None) is None:
        bucket_of, n_groups = bucket_map(tree, B)
        upper = torch.nonzero(bucket_of >= n_groups).view(-1).to(torch.int64)
        states = torch.cat([torch.zeros((1,), dtype=torch.int64), upper])
        plan.upper_rows = RowList(torch.cat([states, states + tree.S]), 2 * tree.S, tree.device)
    return plan.upper_rows


class Stage:
    """Workspace of a staged actor's second level (rnad_bucket_stage_*): per lane the group subtree root it enters, two stamp tables over
    the states, and the two row lists (LiveRows-shaped: rows int32 [2S], count = a view of the workspace's first two int64)."""

    def __init__(self, tree, B, device):
        n = int(lib().rnad_bucket_stage_bytes(tree.ptr, B))
        self.buf = torch.zeros(((n + 7) // 8,), dtype=torch.int64, device=device)
        self.lists = []
        for level in range(2):
            rows = RowList.__new__(RowList)
            rows.N = 2 * tree.S
            rows.rows = torch.empty((2 * tree.S,), dtype=I32, device=device)
            rows.count = self.buf[level: level + 1]
            self.lists.append(rows)


def bucket_stage(tree, B):
    """The Stage of (tree, B), cached on the plan (its buffers keep their addresses: a captured graph of the step holds them)."""
    plan = bucket_plan(tree, B)
    if getattr(plan, "stage", None) is None:
        if int(lib().rnad_bucket_stage_bytes(tree.ptr, B)) < 0:
            return None  # (more than 2^26 states: one staging level)
        plan.stage = Stage(tree, B, tree.device)
    return plan.stage


def bucket_stage_rows(tree, B, stage, level, seed=0, step_params=None):
    """rnad_bucket_stage_rows: the row list of staging level 0 (the group subtree roots the lanes enter; after bucket_sort(stage=...)) or 1
    (the subtrees below the states the lanes were drawn into; after bucket_stage_walk).  Same seed / step_params as the sort."""
    rows = stage.lists[level]
    if level == 0:
        return rows  # (bucket_sort(stage=...) had its last kernel write this list: no launch here)
    _check(lib().rnad_bucket_stage_rows(tree.ptr, B, level, seed, _dp(step_params, torch.int64, "step_params", True),
                                        _dp(stage.buf, torch.int64, "stage"), _dp(rows.rows, I32, "rows"), _stream()))
    return rows


def bucket_stage_walk(tree, traj, buckets, stage, policy_rows, seed=0, lane0=0, step_params=None):
    """rnad_bucket_stage_walk: every lane's transition at the root of the group subtree it enters, drawn as bucket_play will draw it from
    `policy_rows` (the actor's policy rows [2S, stride], evaluated on bucket_stage_rows(level 0)); stamps the states the lanes land in."""
    assert policy_rows.shape[0] == 2 * tree.S
    _check(lib().rnad_bucket_stage_walk(tree.ptr, traj.T_cap, traj.B, _dp(policy_rows, F32, "policy_rows"), policy_rows.shape[1], seed, lane0,
                                        _dp(step_params, torch.int64, "step_params", True), _dp(buckets.plan.scratch, I32, "scratch"),
                                        _dp(buckets.lane_ids, I32, "lane_ids"), _dp(stage.buf, torch.int64, "stage"), _stream()))


def bucket_sort(tree, traj, table, seed=0, lane0=0, step_params=None, table_is_policy=False, column=0, want_flags=False, want_rows=True,
                visited=None, stage=None):
    """rnad_bucket_sort: the first half of rollout_bucketed_compact (keys + sort) for an actor evaluated in stages; `table` needs the
    rows of bucket_upper_rows() only.  Returns (buckets, rows, flags): rows = a RowList of both players' rows of every state inside a
    group the batch descends into (written by the sort's last kernel: evaluate the actor on it, then call bucket_play); flags (want_flags)
    the same set as int32 [2S] marks.  visited (int32 [2S]): cleared here for bucket_play(visited_is_clear=True).
    stage (bucket_stage()): the keys pass also records what the second staging level needs (bucket_stage_rows / bucket_stage_walk)."""
    assert traj.compact and traj.T_cap <= COMPACT_MAX_STEPS
    plan = bucket_plan(tree, traj.B)
    if plan is None:
        raise RnadHipError(lib().rnad_last_error().decode())
    assert table.shape[0] == 2 * tree.S and table.shape[1] >= column + tree.A
    buckets = Buckets(plan, traj.device)
    flags = torch.empty((2 * tree.S,), dtype=I32, device=traj.device) if want_flags else None
    rows = None
    if want_rows:
        rows = RowList.__new__(RowList)
        rows.N = 2 * tree.S
        rows.rows = torch.empty((2 * tree.S,), dtype=I32, device=traj.device)
        rows.count = torch.empty((1,), dtype=torch.int64, device=traj.device)
    base = _dp(table, F32, "table")
    _check(lib().rnad_bucket_sort(tree.ptr, traj.T_cap, traj.B, C.c_void_p(base.value + 4 * column), table.shape[1], int(table_is_policy), seed,
                                  lane0, _dp(step_params, torch.int64, "step_params", True), _dp(plan.scratch, I32, "scratch"),
                                  _dp(buckets.lane_ids, I32, "lane_ids"), _dp(buckets.items, I32, "items"), _dp(buckets.n_items, I32, "n_items"),
                                  _dp(buckets.norm, F64, "norm"), _dp(flags, I32, "group_flags", True),
                                  _dp(rows.rows, I32, "staged_rows") if rows is not None else None,
                                  C.c_void_p(rows.count.data_ptr()) if rows is not None else None, _dp(visited, I32, "visited", True),
                                  _dp(stage.buf, torch.int64, "stage") if stage is not None else None,
                                  _dp(stage.lists[0].rows, I32, "stage_rows0") if stage is not None else None, _stream()))
    return buckets, rows, flags


def bucket_play(tree, traj, buckets, table, rows=None, seed=0, lane0=0, step_params=None, table_is_policy=False, column=0, visited=None,
                defer_alive=False, visited_is_clear=False):
    """rnad_bucket_play: the second half (the rollout in bucket order + alive counts); same seed / lane0 / step_params as bucket_sort.
    rows: the LiveRows the actor was evaluated on since the sort (a logits table: their policy head is taken here).
    visited_is_clear: bucket_sort(visited=...) cleared the flags already (no launch for it here)."""
    assert table.shape[0] == 2 * tree.S and table.shape[1] >= column + tree.A
    _complete_pending(tree, buckets.plan, buckets)
    base = _dp(table, F32, "table")
    _check(lib().rnad_bucket_play(tree.ptr, traj.T_cap, traj.B, C.c_void_p(base.value + 4 * column), table.shape[1], int(table_is_policy),
                                  *_row_list(rows), seed, lane0, _dp(step_params, torch.int64, "step_params", True),
                                  _dp(buckets.plan.scratch, I32, "scratch"), _dp(buckets.lane_ids, I32, "lane_ids"),
                                  _dp(buckets.items, I32, "items"), _dp(buckets.n_items, I32, "n_items"),
                                  _dp(buckets.norm, F64, "norm"), _dp(traj.states, traj.states.dtype, "states"),
                                  None if defer_alive else _dp(traj.alive, I32, "alive"), _dp(traj.acts, torch.int64, "acts"),
                                  _dp(traj.final_reward, F32, "final_reward"), _dp(visited, I32, "visited", True), int(bool(visited_is_clear)),
                                  _stream()))
    buckets.alive_pending = traj if defer_alive else None
    buckets.plan._pending = buckets if defer_alive else None
    traj._owner = (tree, buckets)
    traj.invalidate()


def bucket_alive(tree, buckets):
    """rnad_bucket_alive: completes a rollout_bucketed_compact(defer_alive=True) -- traj.alive and buckets.norm -- on its own."""
    traj = buckets.alive_pending
    if traj is None:
        return
    _check(lib().rnad_bucket_alive(tree.ptr, traj.T_cap, traj.B, _dp(buckets.plan.scratch, I32, "scratch"), _dp(traj.alive, I32, "alive"),
                                   _dp(buckets.norm, F64, "norm"), _stream()))
    buckets.alive_pending = None


def bucket_indices(tree, buckets, traj):
    """rnad_bucket_indices: the reference's indices (int32 [T_cap + 1, B], episode.py:218) of a compact trajectory -- the bucket's own
    path states above the cut, bucket_lo + relative state below."""
    out = torch.empty((traj.T_cap + 1, traj.B), dtype=I32, device=traj.device)
    _check(lib().rnad_bucket_indices(tree.ptr, traj.T_cap + 1, traj.B, _dp(traj.states, traj.states.dtype, "states"),
                                     _dp(buckets.items, I32, "items"), _dp(buckets.n_items, I32, "n_items"), _dp(out, I32, "indices"), _stream()))
    return out


def bucket_pack_states(tree, buckets, traj, indices):
    """rnad_bucket_pack_states: a bucket-ordered int32 [T_cap + 1, B] trajectory (column j = lane buckets.lane_ids[j]) into the relative
    states of the compact Trajectory `traj`.  Raises if a column does not belong to its work item's bucket."""
    assert traj.compact and tuple(indices.shape) == (traj.T_cap + 1, traj.B)
    bad = torch.zeros((1,), dtype=I32, device=traj.device)
    _check(lib().rnad_bucket_pack_states(tree.ptr, traj.T_cap + 1, traj.B, _dp(indices.contiguous(), I32, "indices"), _dp(buckets.items, I32, "items"),
                                         _dp(buckets.n_items, I32, "n_items"), _dp(traj.states, traj.states.dtype, "states"),
                                         _dp(bad, I32, "mismatch"), _stream()))
    if int(bad.item()):
        raise RnadHipError("bucket_pack_states: a column of `indices` is not a lane of its work item's bucket")
    traj._owner = (tree, buckets)
    traj.invalidate()


def bucket_expand(tree, traj, records):
    """rnad_bucket_expand: the dense mask_bits / policy / actions / rewards [T_cap, B] buffers of a compact trajectory."""
    dev, T, B, A = traj.device, traj.T_cap, traj.B, tree.A
    traj.mask_bits = torch.empty((T, B), dtype=U8, device=dev)
    traj.policy = torch.empty((T, B, A), dtype=F32, device=dev)
    traj.actions = torch.empty((T, B), dtype=I32, device=dev)
    traj.rewards = torch.empty((T, B), dtype=F32, device=dev)
    _check(lib().rnad_bucket_expand(tree.ptr, T, B, _dp(traj.indices, I32, "indices"), _dp(traj.acts, torch.int64, "acts"),
                                    _dp(traj.final_reward, F32, "final_reward"), _dp(records, F32, "records"),
                                    _dp(traj.mask_bits, U8, "mask_bits"), _dp(traj.policy, F32, "policy"), _dp(traj.actions, I32, "actions"),
                                    _dp(traj.rewards, F32, "rewards"), _stream()))


def learn_bucketed_compact(tree, buckets, traj, T, records, fast_records, norm, hp, want_losses=False, rows=None, groups=None):
    """rnad_learn_bucketed_compact on the first T steps of a compact trajectory played with the pi columns of `records`;
    (records, fast_records) = bucket_records(..., fast=True).  rows: a LiveRows over the 2S rows -- only those rows of the gradient
    tables are written (the batch visited no others).  groups: see bucket_finish."""
    B, A = traj.B, tree.A
    assert buckets.plan.B == B and traj.compact and 1 <= T <= traj.T_cap
    rows, groups = _rows_and_groups(tree, buckets, rows, groups)
    assert rows is None or rows.N == 2 * tree.S
    dev = traj.device
    dlogit = torch.empty((2 * tree.S, A), dtype=F32, device=dev)
    dv = torch.empty((2 * tree.S, 1), dtype=F32, device=dev)
    losses = torch.empty((2,), dtype=F64, device=dev) if want_losses else None
    # the rollout left its alive counts to this launch (defer_alive): its first T_cap + 1 workgroups add them up
    pending = (None, 0, None, None)
    if getattr(buckets, "alive_pending", None) is traj:
        pending = (_dp(buckets.plan.scratch, I32, "scratch"), traj.T_cap, _dp(traj.alive, I32, "alive"), _dp(buckets.norm, F64, "norm"))
    _check(lib().rnad_learn_bucketed_compact(tree.ptr, T, B, _dp(traj.states, traj.states.dtype, "states"), _dp(traj.acts, torch.int64, "acts"),
                                             _dp(traj.final_reward, F32, "final_reward"), _dp(fast_records, F32, "fast_records"),
                                             _dp(records, F32, "records"), _dp(buckets.items, I32, "items"),
                                             _dp(buckets.n_items, I32, "n_items"), _dp(norm, F64, "norm", True), C.byref(hp),
                                             _dp(buckets.plan.accumulators, torch.int64, "accumulators"),
                                             _dp(losses, F64, "losses", True), _dp(dlogit, F32, "dlogit_tab"), _dp(dv, F32, "dv_tab"),
                                             *_row_list(rows), *pending, groups, _stream()))
    buckets.alive_pending = None
    return dlogit, dv, losses


def _row_list(rows):
    """(rows, n_rows) pointer arguments of a LiveRows, or two NULLs."""
    if rows is None:
        return None, None
    return _dp(rows.rows, I32, "rows"), C.c_void_p(rows.count.data_ptr())


def step_params_set(step_params, seed, alpha):
    """step_params (device int64 [2] = struct rnad_step_params) <- seed, alpha, 1 - alpha; enqueued on the current stream."""
    _check(lib().rnad_step_params_set(_dp(step_params, torch.int64, "step_params"), int(seed) & 0xFFFFFFFFFFFFFFFF, float(alpha),
                                      1.0 - float(alpha), _stream()))


STEP_QUEUE = 32  # RNAD_STEP_QUEUE
STEP_QUEUE_WORDS = 2 + 2 + 2 * STEP_QUEUE  # int64 words of a struct rnad_step_queue


class StepParams(C.Structure):
    """struct rnad_step_params (include/rnad_hip.h)."""

    _fields_ = [("seed", C.c_uint64), ("alpha", C.c_float), ("one_minus_alpha", C.c_float)]


def step_entry(seed, alpha):
    """(seed, alpha, 1 - alpha) as the device will hold them: the two floats rounded to fp32 the way step_params_set passes them."""
    e = StepParams(int(seed) & 0xFFFFFFFFFFFFFFFF, float(alpha), 1.0 - float(alpha))
    return (e.seed, e.alpha, e.one_minus_alpha)


def step_queue_set(queue, entries):
    """queue (device int64 [STEP_QUEUE_WORDS] = struct rnad_step_queue) <- the scalars of the next len(entries) steps, entries =
    [step_entry(seed, alpha), ...]; its first 16 bytes are the struct rnad_step_params the kernels read (pass `queue` as step_params),
    and OptimizerStep(..., advance=queue) moves on to the next entry at the end of every step."""
    n = len(entries)
    assert 1 <= n <= STEP_QUEUE and queue.numel() >= STEP_QUEUE_WORDS
    arr = (StepParams * n)(*[StepParams(*e) for e in entries])
    _check(lib().rnad_step_queue_set(_dp(queue, torch.int64, "step_queue"), n, arr, _stream()))


def policy_column(A):
    """First column of the learner's policy pi[A] inside a bucket_records row."""
    return 3 * A + 3


def bucket_records(tree, logit_tab, v_tab, v_target_tab, logit_reg_tab, logit_reg_tab_, hp, step_params=None, fast=False, rows=None):
    """One record per (player, state) row with everything of the update that depends on the row alone (rnad_bucket_records):
    logit[A] | v | v_target | process_policy(pi)[A] | log_policy_reg[A] | legal bits | pi[A] | pad.
    fast=True: returns (records, fast_records) -- the second table holds the row-only operands of the on-policy learner
    (learn_bucketed_compact; layout in include/rnad_hip.h).  rows: a LiveRows over the 2S rows -- only those records are written."""
    stride = int(lib().rnad_bucket_record_stride(tree.A))
    assert rows is None or rows.N == 2 * tree.S
    rec = torch.empty((2 * tree.S, stride), dtype=F32, device=logit_tab.device)
    quick = torch.empty((2 * tree.S, int(lib().rnad_bucket_fast_record_stride(tree.A))), dtype=F32, device=logit_tab.device) if fast else None
    # the actor's policy rows on their own (16 bytes per row at A <= 4): what the rollout kernels gather from.  Only a full table can
    # be an actor, so a row-list call does without
    pol = (torch.empty((2 * tree.S, int(lib().rnad_bucket_policy_row_stride(tree.A))), dtype=F32, device=logit_tab.device)
           if (fast and rows is None and os.environ.get("RNAD_POLICY_ROWS", "1") == "1") else None)
    _check(lib().rnad_bucket_records(tree.ptr, _dp(logit_tab, F32, "logit_tab"), _dp(v_tab, F32, "v_tab"), _dp(v_target_tab, F32, "v_target_tab"),
                                     _dp(logit_reg_tab, F32, "logit_reg_tab"), _dp(logit_reg_tab_, F32, "logit_reg_tab_"), C.byref(hp),
                                     _dp(step_params, torch.int64, "step_params", True), _dp(rec, F32, "records"),
                                     _dp(quick, F32, "fast_records", True), _dp(pol, F32, "policy_rows", True), *_row_list(rows), _stream()))
    if pol is not None:
        rec._policy_rows = pol  # travels with the records: rollout_bucketed_compact(table=records) gathers from it
    return (rec, quick) if fast else rec


def complete_records(records):
    """Distinct observations: the default step copies a representative's fast record and policy row to the other rows of its group (the
    rollout and the learner gather those per row) but leaves the 64-byte ROW records of the other rows to whoever reads them -- the dense
    views of a compact batch (bucket_expand), the dense-trajectory learner.  This makes the copies if they are still owed (`_expand_job` /
    `_expand_stale`, set where the records are written: learn/rnad.py _table_outputs, Episodes.invalidate_derived after a graph replay)."""
    job = getattr(records, "_expand_job", None)
    if job is not None and getattr(records, "_expand_stale", False):
        rows_expand(*job)
        records._expand_stale = False


def rows_expand(dedup, tables):
    """rnad_rows_expand: every table (float32 [2S, k], k a multiple of 4) gets, in the rows that are not representatives, the row of their
    representative (dedup: TreeHandle.obs_dedup())."""
    tables = [t for t in tables if t is not None]
    assert 1 <= len(tables) <= 4 and all(t.shape[0] >= dedup.n_rows and t.dtype == F32 for t in tables)
    ptrs = (C.c_void_p * len(tables))(*[_dp(t, F32, "table").value for t in tables])
    widths = (C.c_int32 * len(tables))(*[int(t.shape[1]) for t in tables])
    _check(lib().rnad_rows_expand(C.c_int64(dedup.n_rows), _dp(dedup.rep_of, I32, "rep_of"), len(tables), ptrs, widths, _stream()))


def rows_segment_sum(dedup, A, dlogit_tab, dv_tab):
    """rnad_rows_segment_sum: the representatives' rows of dlogit_tab [2S, A] / dv_tab [2S, 1] receive the sums over their groups."""
    _check(lib().rnad_rows_segment_sum(dedup.n_multi, _dp(dedup.multi_start, I32, "start"), _dp(dedup.multi_order, I32, "order"), A,
                                       _dp(dlogit_tab, F32, "dlogit_tab"), _dp(dv_tab, F32, "dv_tab"), _stream()))


def mlp_rows_records_supported(A, W, fold=False, from_table=False):
    return os.environ.get("RNAD_FUSED_ROWS", "1") == "1" and bool(lib().rnad_mlp_rows_records_supported(A, W, int(bool(fold)), int(bool(from_table))))


def mlp_rows_records(tree, packed_net, packed_target, W, obs, logit_reg_tab, logit_reg_tab_, hp, step_params=None, fold=False, rows=None,
                     logit_tab=None, want_records=True, want_policy_rows=True, alloc_rows=None):
    """rnad_mlp_rows_records: learner (both heads) and target (value head) on the tree's observation table `obs` AND the row records of
    bucket_records(fast=True), in one launch.  Returns dict(logit [2S, A], v [2S, 1], v_target [2S, 1], records, fast_records); the
    actor's policy rows travel as records._policy_rows, as with bucket_records.
    logit_tab: the learner's logits already exist (a staged actor wrote them) -- only the value heads are evaluated; rows: a LiveRows
    over the 2S rows -- only those rows are evaluated and written (policy rows are then not produced: only a full table can be an actor)."""
    fold = _fold_checked(fold, obs, "mlp_rows_records")
    A, N = tree.A, 2 * tree.S
    assert obs.numel() == N * 2 * A * A, "mlp_rows_records: obs must be the tree's observation table"
    half = obs.dtype == F16
    dev = obs.device
    assert rows is None or rows.N == N
    from_table = logit_tab is not None
    # alloc_rows (>= 2S): rows of the output tables -- a caller that all-gathers row shards of equal size pads them (rows beyond 2S are
    # never written); with it the policy rows are produced for a row list too
    M = N if alloc_rows is None else int(alloc_rows)
    assert M >= N
    logit = logit_tab if from_table else torch.empty((M, A), dtype=F32, device=dev)
    v = torch.empty((M, 1), dtype=F32, device=dev)
    vt = torch.empty((M, 1), dtype=F32, device=dev)
    rec = torch.empty((M, int(lib().rnad_bucket_record_stride(A))), dtype=F32, device=dev) if want_records else None
    quick = torch.empty((M, int(lib().rnad_bucket_fast_record_stride(A))), dtype=F32, device=dev)
    pol = (torch.empty((M, int(lib().rnad_bucket_policy_row_stride(A))), dtype=F32, device=dev)
           if (want_policy_rows and (rows is None or alloc_rows is not None) and not from_table and os.environ.get("RNAD_POLICY_ROWS", "1") == "1")
           else None)
    _check(lib().rnad_mlp_rows_records(tree.ptr, W, int(fold), _dp(packed_net, F32, "packed_net"), _dp(packed_target, F32, "packed_target"),
                                       _dp(obs, F16 if half else F32, "obs"), int(half), *_row_list(rows), int(from_table),
                                       _dp(logit, F32, "logit_tab"), _dp(v, F32, "v_tab"), _dp(vt, F32, "v_target_tab"),
                                       _dp(logit_reg_tab, F32, "logit_reg_tab"), _dp(logit_reg_tab_, F32, "logit_reg_tab_"), C.byref(hp),
                                       _dp(step_params, torch.int64, "step_params", True), _dp(rec, F32, "records", True),
                                       _dp(quick, F32, "fast_records"), _dp(pol, F32, "policy_rows", True), _stream()))
    if pol is not None and rec is not None:
        rec._policy_rows = pol
    return dict(logit=logit, v=v, v_target=vt, records=rec, fast_records=quick, policy_rows=pol)


def learn_bucketed(tree, buckets, indices, actions, rewards, mu, records, norm, hp, want_losses=False):
    """rnad_learn_bucketed on a bucket-ordered trajectory -> dlogit_tab [2S, A], dv_tab [2S, 1], losses f64[2] or None."""
    T, B, A = mu.shape
    assert buckets.plan.B == B
    dev = mu.device
    dlogit = torch.empty((2 * tree.S, A), dtype=F32, device=dev)
    dv = torch.empty((2 * tree.S, 1), dtype=F32, device=dev)
    losses = torch.empty((2,), dtype=F64, device=dev) if want_losses else None
    _check(lib().rnad_learn_bucketed(tree.ptr, T, B, _dp(indices, I32, "indices"), _dp(actions, I32, "actions"), _dp(rewards, F32, "rewards"),
                                     _dp(mu, F32, "mu"), _dp(records, F32, "records"), _dp(buckets.items, I32, "items"),
                                     _dp(buckets.n_items, I32, "n_items"), _dp(norm, F64, "norm", True), C.byref(hp),
                                     _dp(buckets.plan.accumulators, torch.int64, "accumulators"), _dp(losses, F64, "losses", True),
                                     _dp(dlogit, F32, "dlogit_tab"), _dp(dv, F32, "dv_tab"), _stream()))
    return dlogit, dv, losses


def bucket_finish(tree, buckets, norm, hp, dlogit, dv, losses=None, rows=None, groups=None):
    """rnad_bucket_finish: completes a learn_bucketed / learn_bucketed_compact call that was made with norm=None.  groups: an ObsDedup
    with groups_below_cut(tree, buckets.plan) -- rows is then its `singles`, and the groups' sums end up in their representatives' rows."""
    rows, groups = _rows_and_groups(tree, buckets, rows, groups)
    _check(lib().rnad_bucket_finish(tree.ptr, buckets.plan.B, _dp(norm, F64, "norm"), C.byref(hp),
                                    _dp(buckets.plan.accumulators, torch.int64, "accumulators"), _dp(losses, F64, "losses", True),
                                    _dp(dlogit, F32, "dlogit_tab"), _dp(dv, F32, "dv_tab"), *_row_list(rows), groups, _stream()))


def _rows_and_groups(tree, buckets, rows, groups):
    """(rows, pointer to the rnad_row_groups_t) of a finish that adds the groups of an ObsDedup up."""
    if groups is None:
        return rows, None
    assert rows is None or rows is groups.singles, "with groups, the row list is the rows outside them"
    assert groups.n_rows == 2 * tree.S and groups.groups_below_cut(tree, buckets.plan)
    # r06: the singles BELOW the cut of this plan (the rows above it are converted by the finish's wave-per-row workgroups whatever the list
    # says): with them listed apart the row threads need no per-row lookup of the row's bucket.  Cached per batch size.
    below = groups.__dict__.setdefault("_singles_below", {})
    if buckets.plan.B not in below:
        bucket_of, n_groups = bucket_map(tree, buckets.plan.B)
        dev = groups.singles.rows.device
        upper = (bucket_of >= n_groups).to(dev)
        r = groups.singles.rows[: int(groups.singles.count.item())].long()
        keep = ~upper[torch.where(r >= tree.S, r - tree.S, r)]
        lst = RowList(r[keep].to(I32), 2 * tree.S, dev)
        c = RowGroups(groups.c_groups.n_groups, groups.c_groups.start, groups.c_groups.order, groups.c_groups.first, 1)
        _publish_shared(dev)
        below[buckets.plan.B] = (lst, c)
    lst, c = below[buckets.plan.B]
    return lst, C.byref(c)


def clip_grad_norm(flat, max_norm):
    """In-place clip_grad_norm_ of one flat fp32 gradient bucket (rnad_clip_grad_norm)."""
    _check(lib().rnad_clip_grad_norm(C.c_int64(flat.numel()), _dp(flat, F32, "grads"), C.c_float(max_norm), None, _stream()))


class AdamParams(C.Structure):
    """struct rnad_adam_params (include/rnad_hip.h)."""

    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("max_norm", C.c_float), ("ema", C.c_float)]


class OptimizerStep:
    """rnad_optimizer_step bound to fixed tensors: clip + Adam + EMA of `params` (gradients back to back in one flat bucket, in that
    order) in one launch, on torch.optim.Adam's own state tensors.  The pointer arrays are built once."""

    def __init__(self, params, exp_avg, exp_avg_sq, steps, targets, lr, beta1, beta2, eps, max_norm, ema, packed=None, A=0, fold=False):
        """packed = (image of params, image of targets) with A: the tensors are the fused MLP's (MLP_KEYS order) and the kernel also
        keeps those two packed weight images (mlp_pack) current -- no pack launch per step."""
        n = len(params)
        assert 1 <= n <= 8 and len(exp_avg) == len(exp_avg_sq) == len(steps) == n and (targets is None or len(targets) == n)
        self._keep = (params, exp_avg, exp_avg_sq, steps, targets, packed)
        self.packed, self.A, self.W = packed, (int(A) if packed is not None else 0), (params[0].shape[0] if packed is not None else 0)
        self.fold = int(bool(fold) and packed is not None)  # the images are in the FOLD layout
        self.n = n
        self.numel = sum(p.numel() for p in params)
        arr = lambda ts, name: (C.c_void_p * n)(*[_dp(t, F32, name).value for t in ts])  # noqa: E731
        self.sizes = (C.c_int64 * n)(*[p.numel() for p in params])
        self.param, self.m, self.v, self.step = arr(params, "param"), arr(exp_avg, "exp_avg"), arr(exp_avg_sq, "exp_avg_sq"), arr(steps, "step")
        self.target = arr(targets, "target") if targets is not None else None
        self.hp = AdamParams(float(lr), float(beta1), float(beta2), float(eps), float(max_norm), float(ema))
        # the kernel's workgroups count themselves here (and the last one resets it): this optimiser's own word, not a device global
        self.ticket = torch.zeros((1,), dtype=I32, device=params[0].device)

    def __call__(self, flat, advance=None):
        """advance: a step queue (step_queue_set) to move on once the step is over."""
        assert flat.numel() == self.numel and (advance is None or advance.numel() >= STEP_QUEUE_WORDS)
        img = self.packed or (None, None)
        _check(lib().rnad_optimizer_step(self.n, self.sizes, self.param, _dp(flat, F32, "grads"), self.m, self.v, self.step, self.target,
                                         C.byref(self.hp), None, self.A, self.W, self.fold, _dp(img[0], F32, "packed_param", True),
                                         _dp(img[1], F32, "packed_target", True), _dp(advance, torch.int64, "advance", True),
                                         _dp(self.ticket, I32, "ticket"), _stream()))


def make_learn_params(alpha, eta, lambda_=1.0, c=1.0, rho=1.0, gamma=1.0, clip=1e3, threshold=2.0, w_v=1.0, w_n=1.0,
                      eps_threshold=0.03, n_disc=32):
    # 1 - alpha is taken in double like the reference's python scalar (rnad.py:382), then rounded to fp32
    return LearnParams(float(alpha), 1.0 - float(alpha), float(eta), float(lambda_), float(c), float(rho), float(gamma),
                       float(clip), float(threshold), float(w_v), float(w_n), float(eps_threshold), int(n_disc))


# --------------------------------------------------------------------------------------- NashConv
def nashconv(tree, joint_policy, root_policy, state_index, reach, row_best, col_best, reach_out, depth_out):
    _check(lib().rnad_nashconv(tree.ptr, _dp(joint_policy, F32, "joint_policy"), _dp(root_policy, F32, "root_policy"),
                               C.c_int64(state_index), C.c_float(reach), _dp(row_best, F32, "row_best"),
                               _dp(col_best, F32, "col_best"), _dp(reach_out, F32, "reach_probability"),
                               _dp(depth_out, I32, "depth"), _stream()))


# --------------------------------------------------------------------------------------- host-side: generator / solver
def solve_matrix(M, max_actions):
    """tree.py:199-234 for one fp32 matrix (CPU tensor [ra, ca]) -> (solution [2*max_actions], value)."""
    M = M.detach().to("cpu", F32).contiguous()
    ra, ca = M.shape
    sol = torch.zeros((2 * max_actions,), dtype=F32)
    val = C.c_float()
    _check(lib().rnad_solve_matrix(C.c_void_p(M.data_ptr()), ra, ca, max_actions, C.c_void_p(sol.data_ptr()), C.byref(val)))
    return sol, val.value


def tree_generate(A, Cc, depth_bound, transition_threshold=0.0, terminal_values=(-1.0, 1.0), prune=(0, 0), seed=0):
    """Native regular-tree generator (rnad_tree_generate).  Returns a dict of CPU tensors in the reference layout."""
    tv = torch.tensor(list(terminal_values), dtype=F32)
    args = (A, Cc, int(depth_bound), C.c_float(transition_threshold), C.c_void_p(tv.data_ptr()), tv.numel(), int(prune[0]),
            int(prune[1]), C.c_uint64(seed))
    null = C.c_void_p()
    S = lib().rnad_tree_generate(*args, C.c_int64(0), null, null, null, null, null, null, null)
    if S < 0:
        _check(1)
    out = dict(
        index=torch.zeros((S, Cc, A, A), dtype=torch.int64), value=torch.zeros((S, Cc, A, A), dtype=F32),
        chance=torch.zeros((S, Cc, A, A), dtype=F32), expected_value=torch.zeros((S, 1, A, A), dtype=F32),
        legal=torch.zeros((S, 1, A, A), dtype=F32), root_value=torch.zeros((S, 1), dtype=F32),
        solution=torch.zeros((S, 2 * A), dtype=F32),
    )
    S2 = lib().rnad_tree_generate(*args, C.c_int64(S), *[C.c_void_p(out[k].data_ptr()) for k in
                                                         ("index", "value", "chance", "expected_value", "legal", "root_value", "solution")])
    if S2 != S:
        _check(1 if S2 < 0 else 0)
        raise RnadHipError(f"rnad_tree_generate is not deterministic: {S} then {S2} states")
    return out


# --------------------------------------------------------------------------------------- profiling hooks
PROF_OBSERVE, PROF_ACT, PROF_LEARN, PROF_MLP, PROF_MLP_BWD = 0, 1, 2, 3, 4
PROF_BUCKET_KEYS, PROF_BUCKET_SORT, PROF_BUCKET_ROLLOUT, PROF_BUCKET_LEARN, PROF_BUCKET_FINISH = 5, 6, 7, 8, 9


def source_hash():
    """sha256 (first 16 hex digits) over the sources librnad_hip.so is built from (csrc/*.hip, *.hpp, *.cpp, Makefile, include/*.h):
    what a measurement that is kept in a file (rocprofv3 counter passes, profiles/) names as the build it was taken on."""
    import glob
    import hashlib

    csrc = os.path.join(_HERE, "..", "csrc")
    inc = os.path.join(_HERE, "..", "..", "include")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.cpp"))
                   + [os.path.join(csrc, "Makefile")] + glob.glob(os.path.join(inc, "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def prof_enable(on):
    """True: bracket every kernel; False: off; or an iterable of PROF_* ids to bracket only those."""
    if on is True:
        mask = -1
    elif not on:
        mask = 0
    else:
        mask = sum(1 << int(k) for k in on)
    _check(lib().rnad_prof_enable(mask))


def prof_read(which):
    n, ms = C.c_int64(), C.c_double()
    _check(lib().rnad_prof_read(int(which), C.byref(n), C.byref(ms)))
    return n.value, ms.value
