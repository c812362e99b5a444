"""Host-side mirror of the reference's ``algoParams`` (param.hpp:4-31, defaults :18-31)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class RaftParams:
    reso: int = 50                 # -r
    est_cov: int = 0               # -e (mandatory > 0, main.cpp:65-68)
    cov_mul: float = 1.5           # -m
    repeat_length: int = 10000     # -p (sets interval_length too, main.cpp:44-47)
    interval_length: int = 10000
    read_length: int = 20000       # -l
    overlap_length: int = 500      # -v
    flanking_length: int = 1000    # -f
    symmetric_mode: int = -1       # -1: detect like chop.hpp:175-184; 0/1: asserted by the caller

    @property
    def high_cov(self) -> int:
        """repeat.hpp:89-90: ``int high_cov = cov_est * param.cov_mul`` (int * double, truncated)."""
        return int(int(self.est_cov) * float(self.cov_mul))

    def cli_args(self) -> list[str]:
        """Flags for the reference binary / our ``raft`` CLI that reproduce these values.

        ``-p`` sets repeat_length and interval_length together (main.cpp:44-47); ``-v`` falls
        through into ``-o`` (main.cpp:51-55), so it must come before any ``-o``.
        """
        if self.repeat_length != self.interval_length:
            raise ValueError("the CLI cannot set repeat_length != interval_length")
        return ["-r", str(self.reso), "-e", str(self.est_cov), "-m", repr(float(self.cov_mul)),
                "-l", str(self.read_length), "-p", str(self.repeat_length), "-f", str(self.flanking_length),
                "-v", str(self.overlap_length)]
