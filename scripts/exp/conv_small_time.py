"""fp32 conv forward / adjoint at chosen shapes; FMI_KS=<n> overrides the reduction split"""
import sys
sys.path.insert(0, "/root/repo")
sys.argv = [sys.argv[0], "none"]
import scripts.microbench as mb
for (n, h, c, k) in [(24, 56, 256, 256), (24, 28, 256, 512), (24, 28, 512, 512), (8, 128, 256, 128), (8, 64, 256, 256), (24, 112, 128, 128)]:
    mb.conv_case(n, h, c, k)
