"""whole-pSp fixture (tests/test_gpu_psp.py::test_psp_whole_train_against_reference) twice in the reproducible mode and once in the default
mode: prints the gradient-error distributions against the float64 digests (the reproducible runs print identical lines)"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, pytest
from face_mask_inpaint_amd import functional as FF
import test_gpu_psp as T
from conftest import GOLDEN
cache = {}
def golden(name):
    if name not in cache: cache[name] = torch.load(os.path.join(GOLDEN, name), weights_only=True)
    return cache[name]
dev = torch.device("cuda:0")
for mode in (True, True, False):
    print("==== reproducible" if mode else "==== default", flush=True)
    try:
        T.test_psp_whole_train_against_reference(dev, golden, mode)
    except AssertionError as e:
        print("ASSERT:", str(e)[:400])
