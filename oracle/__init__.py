"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy / CPU PyTorch, plus a C port under oracle/csrc) of the algorithms on
InstanceRefer's hot path. Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import anything from here, and only as the checker / reported CPU baseline — never as the product
path. instancerefer_amd/ must not import this package (tests/test_no_oracle_in_product.py enforces it).

PARITY STATUS: the sparse arithmetic lives in third-party packages that are NOT in the reference tree
and cannot be installed here (no network):
  * torchsparse  — mit-han-lab/torchsparse "1.2" per reference README.md:41, installed unpinned from
    git HEAD (README.md:43); API used = v1.1/v1.2 (`SparseTensor(feats, coords)`, `.F/.C/.s`).
  * torch_geometric / torch_cluster / torch_scatter — unpinned, wheel index torch-1.6.0+cu101
    (reference requirement.txt:9-13).
The reference has no tests, golden vectors or fixtures. So for the torchsparse /
torch_cluster operators this oracle restates their PUBLISHED algorithms and is "parity unpinned";
it is cross-checked by an independent dense formulation (F.conv3d on a densified grid,
tests/test_oracle_dense.py). What IS pinned: the reference's own Python (lang_module, loss_helper,
attribute/relation/scene glue) is imported from /root/reference in the build container and run on
top of these stubs to generate tests/golden/*.npz (tests/golden/make_golden.py).
"""
