"""Dev: resident workgroups per CU as the HIP runtime computes them for the main MFMA kernels."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instancerefer_amd import _lib
torch.zeros(1, device="cuda")
L = ctypes.CDLL(_lib.load()._name)
for i, n in enumerate(("k_spconv2<128,128> fp32", "k_spconv2<64,64> fp32", "k_spconv2<128,128> bf16 storage", "k_spconv2<64,64> bf16 storage")):
    print("%-36s %d workgroups of 256 threads per CU" % (n, L.irx_debug_occupancy(i)))
for i, n in enumerate(("k_wgrad_pairs<128,128> fp32", "k_wgrad_pairs<64,64> fp32", "k_wgrad_pairs<128,128> bf16 storage")):
    print("%-36s %d workgroups of 256 threads per CU" % (n, L.irx_debug_occupancy_wp(i)))
for i, n in enumerate(("k_spconv3<128,128,4,KH=2>", "k_spconv3<64,64,4>", "k_spconv3<128,128,4,KH=1>", "k_spconv3<64,128,4>")):
    print("%-36s %d workgroups per CU" % (n, L.irx_debug_occupancy_s3(i)))
