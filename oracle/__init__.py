"""CPU oracle of the PSELDNets hot path — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU / numpy restatement of the reference's algorithm (each function cites the reference
file:line it follows, relative to /root/reference/src). Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package, and only as the checker: the product (pseldnets_amd/) never does.

Pinning: the reference is pure Python and ships no tests or golden vectors. The oracle is pinned
by fixtures under tests/golden/ generated in the build container by importing the reference's own modules
(tests/golden/make_golden.py). The STFT / mel / dB arithmetic lives in torchaudio 2.2.1, which is absent from
/root/reference and from this image: that part is restated from torchaudio's published algorithm and
cross-checked against an independent float64 numpy restatement — "parity unpinned" for torchaudio's own
arithmetic, pinned for everything the reference implements itself (intensity vector, HTS-AT, heads, losses).
"""
