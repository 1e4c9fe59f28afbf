"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference U-Net hot path (BoySun045/Pytorch-UNet).  Nothing in the
product package (`pytorch-unet_amd/`) imports this directory; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may.

Parity pinning: the reference ships no tests and no golden vectors, so
this oracle is pinned against outputs of the reference itself, generated in the build
container by `oracle/make_golden.py` (imports `/root/reference/unet` and
`/root/reference/utils/dice_score.py` on CPU) and committed under `tests/golden/`.
"""
