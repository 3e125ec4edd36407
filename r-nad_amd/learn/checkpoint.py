"""On-disk layout of an R-NaD run -- the FORMAT is the reference's (learn/rnad.py:200-319), so runs written by either side
resume on the other:

    <root>/saved_runs/<name>/params      torch.save of {constructor member: value}  (reference `saved_keys`, rnad.py:153,208)
    <root>/saved_runs/<name>/<m>/<n>     torch.save of {total_steps, net_params, net, net_target, net_reg, net_reg_, optimizer}
                                         after n learner steps of regularisation update m       (rnad.py:307-319)

Resuming continues from the largest m, and within it the largest n (rnad.py:263-271).  Files are written to a temporary name
and renamed, so a reader never sees a half-written checkpoint (ranks of a data-parallel run share the directory).
"""
import os
from typing import Optional, Tuple

import torch

CHECKPOINT_KEYS = ("total_steps", "net_params", "net", "net_target", "net_reg", "net_reg_", "optimizer")


def _numeric_entries(path, want_dirs):
    out = []
    if os.path.isdir(path):
        for entry in os.scandir(path):
            if entry.name.isdigit() and entry.is_dir() == want_dirs:
                out.append(int(entry.name))
    return sorted(out)


class RunStore:
    def __init__(self, directory):
        self.directory = directory

    # ---------------------------------------------------------------- what is on disk
    def updates(self):
        """Regularisation updates m that have a directory."""
        return _numeric_entries(self.directory, want_dirs=True)

    def steps(self, m):
        """Learner steps n with a checkpoint inside update m."""
        return _numeric_entries(os.path.join(self.directory, str(m)), want_dirs=False)

    def latest(self) -> Optional[Tuple[int, int]]:
        """(m, n) of the checkpoint a resumed run continues from, or None for a fresh directory."""
        for m in reversed(self.updates()):
            steps = self.steps(m)
            if steps:
                return m, steps[-1]
        return None

    # ---------------------------------------------------------------- atomic torch.save
    @staticmethod
    def _write(obj, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.tmp{os.getpid()}"
        torch.save(obj, tmp)
        os.replace(tmp, path)

    def write_params(self, params: dict):
        self._write(params, os.path.join(self.directory, "params"))

    def read_params(self) -> dict:
        return torch.load(os.path.join(self.directory, "params"), weights_only=False)

    def write(self, m, n, payload: dict):
        assert tuple(sorted(payload)) == tuple(sorted(CHECKPOINT_KEYS)), "checkpoint payload does not match the reference's keys"
        self._write(payload, os.path.join(self.directory, str(m), str(n)))

    def read(self, m, n, map_location=None) -> dict:
        return torch.load(os.path.join(self.directory, str(m), str(n)), map_location=map_location, weights_only=False)
