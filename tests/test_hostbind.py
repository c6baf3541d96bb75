"""Host-thread placement helper (spfsplatv2_amd/hostbind.py): cpulist parsing and the choice of an L3 group on a fake
sysfs tree -- two sockets of two CCDs, as the GPU boxes have them in small."""
from spfsplatv2_amd import hostbind as hb


def _fake_sys(tmp_path):
    groups = {0: "0-3,16-19", 1: "4-7,20-23", 2: "8-11,24-27", 3: "12-15,28-31"}
    for c in range(32):
        d = tmp_path / "devices" / "system" / "cpu" / f"cpu{c}" / "cache" / "index3"
        d.mkdir(parents=True)
        (d / "shared_cpu_list").write_text(groups[(c % 16) // 4] + "\n")
    return tmp_path


def test_cpulists_round_trip():
    assert hb.parse_cpulist("0-7,128-135\n") == list(range(8)) + list(range(128, 136))
    assert hb.parse_cpulist("3") == [3] and hb.parse_cpulist("") == []
    assert hb.format_cpulist([128, 0, 1, 2, 129, 5]) == "0-2,5,128-129"


def test_l3_groups_and_the_deal_among_a_nodes_gpus(tmp_path):
    root = _fake_sys(tmp_path)
    node0 = hb.parse_cpulist("0-7,16-23")                       # socket 0: CCDs 0 and 1 with their SMT siblings
    assert hb.l3_groups(node0, root) == [[0, 1, 2, 3, 16, 17, 18, 19], [4, 5, 6, 7, 20, 21, 22, 23]]
    everything = set(range(32))
    assert hb.choose_group(node0, everything, 0, 2, root) == [0, 1, 2, 3, 16, 17, 18, 19]
    assert hb.choose_group(node0, everything, 1, 2, root) == [4, 5, 6, 7, 20, 21, 22, 23]    # the node's second GPU: the other CCD
    assert hb.choose_group(node0, everything, 0, 1, root) == [0, 1, 2, 3, 16, 17, 18, 19]
    # a cpuset narrower than the node: only what is allowed; nothing local allowed -> no binding
    assert hb.choose_group(node0, {2, 3, 5}, 0, 1, root) == [2, 3]
    assert hb.choose_group(node0, {9, 10}, 0, 1, root) is None
    # unreadable cache topology: single-CPU groups, still a valid choice
    assert hb.choose_group([40, 41], {40, 41}, 1, 2, root) == [41]


def test_binding_without_a_gpu_changes_nothing(tmp_path):
    import os
    before = os.sched_getaffinity(0)
    assert hb.bind_to_gpu_l3(0, tmp_path) is None and os.sched_getaffinity(0) == before
