"""Launch-side host logic of the data-parallel path (faster_rcnn_amd/dp.py) on FAKE sysfs trees: counting the GPUs a child
will see from the KFD topology (no HIP call in a launcher), the *_VISIBLE_DEVICES rules, the CPUs local to a GPU's PCIe
slot, and how the ranks of a node share them out."""
import os

from faster_rcnn_amd import dp


def fake_node(root, n, simd, location_id=0, domain=0):
    d = os.path.join(root, "kfd", str(n))
    os.makedirs(d)
    with open(os.path.join(d, "properties"), "w") as f:
        f.write("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\ndomain %d\nlocation_id %d\n" % (0 if simd else 64, simd, domain, location_id))


def fake_pci(root, bdf, cpulist, numa):
    d = os.path.join(root, "pci", bdf)
    os.makedirs(d)
    open(os.path.join(d, "local_cpulist"), "w").write(cpulist + "\n")
    open(os.path.join(d, "numa_node"), "w").write("%d\n" % numa)


def node_tree(tmp_path, gpus=8):
    """Two CPU nodes, then `gpus` GPUs: the first half on socket 0 (cpus 0-63), the second on socket 1 (64-127)."""
    root = str(tmp_path)
    fake_node(root, 0, 0)
    fake_node(root, 1, 0)
    for g in range(gpus):
        bus = 0x05 + 0x10 * g
        fake_node(root, 2 + g, 1024, location_id=bus << 8)
        fake_pci(root, "0000:%02x:00.0" % bus, "0-63" if g < gpus // 2 else "64-127", 0 if g < gpus // 2 else 1)
    return os.path.join(root, "kfd"), os.path.join(root, "pci")


def test_count_gpus_from_the_kfd_topology(tmp_path):
    kfd, _ = node_tree(tmp_path)
    assert dp.count_gpus(kfd, env={}) == 8                                   # CPU nodes (simd_count 0) are not devices
    assert [n["node"] for n in dp.visible_gpus(kfd, env={})] == [str(i) for i in range(2, 10)]
    assert dp.count_gpus(kfd, env={"HIP_VISIBLE_DEVICES": "0,3"}) == 2
    assert [n["node"] for n in dp.visible_gpus(kfd, env={"HIP_VISIBLE_DEVICES": "3,0"})] == ["5", "2"]      # order is the variable's
    assert dp.count_gpus(kfd, env={"CUDA_VISIBLE_DEVICES": "1"}) == 1
    assert dp.count_gpus(kfd, env={"HIP_VISIBLE_DEVICES": "1", "CUDA_VISIBLE_DEVICES": "0,1,2"}) == 1        # HIP_ wins over its alias
    assert dp.count_gpus(kfd, env={"HIP_VISIBLE_DEVICES": ""}) == 0
    assert dp.count_gpus(kfd, env={"HIP_VISIBLE_DEVICES": "0,9,1"}) == 1     # the runtime stops at the first invalid index
    # ROCR_VISIBLE_DEVICES filters first, HIP_VISIBLE_DEVICES indexes what is left
    assert [n["node"] for n in dp.visible_gpus(kfd, env={"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1"})] == ["7"]
    assert dp.visible_gpus(kfd, env={"ROCR_VISIBLE_DEVICES": "GPU-deadbeef"}) is None                        # UUIDs: ask the runtime
    assert dp.visible_gpus(os.path.join(str(tmp_path), "absent"), env={}) is None


def test_ranks_share_out_the_cores_next_to_their_gpus(tmp_path, monkeypatch):
    kfd, pci = node_tree(tmp_path)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    cpus, numa = dp.gpu_local_cpus(dp.visible_gpus(kfd, env={})[5], pci)
    assert cpus == list(range(64, 128)) and numa == 1
    got = [dp.rank_cpu_slice(r, 8, kfd, pci, env={}) for r in range(8)]
    assert [g[1] for g in got] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert [(g[0][0], g[0][-1], len(g[0])) for g in got] == [(16 * r, 16 * r + 15, 16) for r in range(8)]    # 4 ranks split each socket
    flat = [c for g in got for c in g[0]]
    assert sorted(flat) == list(range(128))                                   # disjoint, nothing left out
    # two ranks on GPUs 0 and 1 (same socket): half a socket each; one rank: left alone
    assert [len(dp.rank_cpu_slice(r, 2, kfd, pci, env={})[0]) for r in range(2)] == [32, 32]
    assert dp.rank_cpu_slice(0, 1, kfd, pci, env={}) == (None, None)
    # a container that may use 8 cores only: 4 ranks on a socket would get 2 each -> below the floor, nobody is pinned
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    assert dp.rank_cpu_slice(0, 8, kfd, pci, env={}) == (None, None)
    # visibility: rank 0 of HIP_VISIBLE_DEVICES=6,7 sits next to socket 1
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    cpus, numa = dp.rank_cpu_slice(0, 2, kfd, pci, env={"HIP_VISIBLE_DEVICES": "6,7"})
    assert numa == 1 and cpus == list(range(64, 96))
    # no PCI information: nothing happens
    assert dp.rank_cpu_slice(0, 8, kfd, os.path.join(str(tmp_path), "nopci"), env={}) == (None, None)


def test_pin_rank_applies_the_slice(tmp_path, monkeypatch):
    kfd, pci = node_tree(tmp_path)
    applied = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: applied.update(pid=pid, cpus=list(cpus)), raising=False)
    msg = dp.pin_rank(5, 8, root=kfd, pci_root=pci, env={})
    assert applied == {"pid": 0, "cpus": list(range(80, 96))} and "80-95" in msg and "numa node 1" in msg
    applied.clear()
    assert dp.pin_rank(0, 1, root=kfd, pci_root=pci, env={}) is None and not applied
