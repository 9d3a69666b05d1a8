"""fp32 conv forward / adjoint at chosen shapes; FMI_KS=<n> overrides the reduction split"""
import sys
sys.path.insert(0, "/root/repo")
sys.argv = [sys.argv[0], "none"]
import scripts.microbench as mb
for (n, h, c, k) in [(16, 32, 256, 256), (16, 64, 128, 128), (16, 16, 512, 512), (16, 128, 64, 64)]:
    mb.conv_case(n, h, c, k)
