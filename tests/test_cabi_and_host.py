"""CPU tests: the C-ABI library loads and exports every symbol include/lynse_hip.h declares, host-side
logic (metric parsing, error mapping, host merge) works without a GPU, and compute entry points fail
loudly (no silent CPU fallback).  No compute calls are made."""
import ctypes as C
import json
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "lynse_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lynse_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    import lynsedb_amd._lib as lb

    syms = declared_symbols()
    assert len(syms) >= 40
    raw = C.CDLL(str(lb.LIB_PATH))
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/lynse_hip.h but not exported"
        assert s in lb.SIGNATURES, f"{s} has no ctypes signature in lynsedb_amd/_lib.py"
    assert set(lb.SIGNATURES) == set(syms)
    assert lb.lib.lynse_hip_abi_version() == 1


def test_library_does_not_link_the_oracle():
    import subprocess

    import lynsedb_amd._lib as lb

    out = subprocess.run(["nm", "-D", str(lb.LIB_PATH)], capture_output=True, text=True).stdout
    assert "lo_" not in "".join(l.split()[-1][:3] for l in out.splitlines() if l.split())
    for src in (ROOT / "lynsedb_amd").rglob("*"):
        if src.suffix in (".py", ".hip", ".h", ".inc", ".cpp") and src.is_file():
            txt = src.read_text(errors="ignore")
            assert "import oracle" not in txt and "from oracle" not in txt and "lynse_oracle" not in txt, src


def test_metric_parsing_matches_reference_and_oracle(oracle, golden_dir):
    import lynsedb_amd as L

    for name in ["ip", "IP", "inner_product", "inner", "dot", "DOT", "l2", "l2sq", "l2_squared", "euclidean",
                 "cosine", "cos", "cosine_distance", "hamming", "jaccard", "dice", "sorensen", "sorensen_dice",
                 "sorensen-dice", "tanimoto"]:
        assert L.metric_from_str(name) == oracle.metric_from_str(name)
    with pytest.raises(ValueError, match="Unknown metric: bogus"):  # src/python/mod.rs:2000-2002
        L.metric_from_str("bogus")
    g = json.loads((golden_dir / "python_reference_vectors.json").read_text())
    for row in g["is_ascending_index"]:  # python/lynse/cluster.py:182 via the golden vectors
        if row["mode"] is None:
            continue
        m = L.metric_from_index_mode(row["mode"])
        assert m == oracle.metric_from_index_mode(row["mode"])
        assert bool(L._lib.lib.lynse_hip_metric_is_ascending(m)) == row["ascending"]
    names = {"IP": 0, "L2": 1, "Cosine": 2, "Hamming": 3, "Jaccard": 4, "Dice": 5, "Tanimoto": 6}
    for row in g["parse_index_mode"]:  # result_view._parse_index_mode
        assert L.metric_from_index_mode(row["mode"]) == names[row["parsed"][1]]
    with pytest.raises(ValueError):
        L.metric_from_index_mode("FLAT-BOGUS")
    assert L._lib.lib.lynse_hip_metric_is_binary(5) == 1 and L._lib.lib.lynse_hip_metric_is_binary(2) == 0


def test_host_merge_matches_golden_and_oracle(oracle, golden_dir):
    import lynsedb_amd as L

    g = json.loads((golden_dir / "python_reference_vectors.json").read_text())
    for case in g["merge_pairs"]:  # cluster._merge_pairs golden outputs
        blocks = case["blocks"]
        stride = max([len(b[0]) for b in blocks] + [1])
        ids = np.zeros((len(blocks), stride), np.uint64)
        ds = np.zeros((len(blocks), stride), np.float32)
        cnt = np.zeros(len(blocks), np.uint32)
        for i, (bi, bs) in enumerate(blocks):
            ids[i, :len(bi)], ds[i, :len(bs)], cnt[i] = bi, bs, len(bi)
        got_i, got_d = L.merge_topk(ids, ds, cnt, case["k"], "l2" if case["ascending"] else "ip")
        assert [int(x) for x in got_i] == case["ids"]
        assert np.allclose(got_d, np.asarray(case["scores"], np.float32))
    rng = np.random.default_rng(1)
    for metric in (0, 1):  # heavy ties: canonical (distance, id) order == oracle merge_results
        ids = rng.permutation(400).astype(np.uint64).reshape(8, 50)
        ds = rng.integers(0, 6, size=(8, 50)).astype(np.float32)
        cnt = rng.integers(0, 51, size=8).astype(np.uint32)
        flat_i = np.concatenate([ids[i, :cnt[i]] for i in range(8)])
        flat_d = np.concatenate([ds[i, :cnt[i]] for i in range(8)])
        e_i, e_d = oracle.merge_results(flat_i, flat_d, 37, metric)
        g_i, g_d = L.merge_topk(ids, ds, cnt, 37, metric)
        assert np.array_equal(e_i, g_i) and np.array_equal(e_d, g_d)


def test_no_cpu_fallback_without_device():
    import lynsedb_amd as L

    if L._lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(L._lib.LynseHipError):
        L.FlatIndex(None, 8)
    with pytest.raises(L._lib.LynseHipError):
        L.py_top_k_search(np.zeros(4, np.float32), np.zeros((3, 4), np.float32), "ip", 2)
    with pytest.raises(ValueError, match="Unknown metric"):  # argument errors are still reported first
        L.py_compute_distance(np.zeros(2, np.float32), np.zeros(2, np.float32), "nope")


def test_block_layout_roundtrip():
    from lynsedb_amd.sharded import ShardedFlat, block_layout

    rng = np.random.default_rng(0)
    nq, k, world = 5, 7, 3
    ro, do, co, total = block_layout(nq, k)
    assert ro == 0 and do == nq * k * 8 and co == do + nq * k * 4 and total % 16 == 0
    blocks, src = [], []
    for r in range(world):
        rows = rng.integers(0, 1 << 40, size=(nq, k)).astype(np.uint64)
        d = rng.random((nq, k), dtype=np.float32)
        c = rng.integers(0, k + 1, size=nq).astype(np.uint32)
        src.append((rows, d, c))
        blocks.append(ShardedFlat.pack_block(rows, d, c))
    R, D, Cn = ShardedFlat.unpack_blocks(np.concatenate(blocks), world, nq, k)
    for r in range(world):
        assert np.array_equal(R[r], src[r][0]) and np.array_equal(D[r], src[r][1]) and np.array_equal(Cn[r], src[r][2])


def test_sift_vecs_reader_matches_golden(golden_dir, tmp_path):
    """benchmarks/sift_io.py read_fvecs/read_ivecs outputs captured from the reference."""
    from lynsedb_amd.datasets import read_fvecs, read_ivecs

    g = json.loads((golden_dir / "python_reference_vectors.json").read_text())
    p = tmp_path / "t.fvecs"
    p.write_bytes(bytes.fromhex(g["fvecs"]["hex"]))
    a = read_fvecs(p)
    assert list(a.shape) == g["fvecs"]["shape"] and a.dtype == np.float32
    assert np.array_equal(a.ravel(), np.asarray(g["fvecs"]["values"], np.float32))
    p = tmp_path / "t.ivecs"
    p.write_bytes(bytes.fromhex(g["ivecs"]["hex"]))
    b = read_ivecs(p)
    assert list(b.shape) == g["ivecs"]["shape"] and np.array_equal(b.ravel(), np.asarray(g["ivecs"]["values"], np.int32))
    with pytest.raises(ValueError):
        (tmp_path / "e.fvecs").write_bytes(b"")
        read_fvecs(tmp_path / "e.fvecs")


def test_hot_kernels_have_no_scratch_spills():
    """The build records every kernel's register / scratch usage; the unfiltered scan kernels of the default path must
    not spill (spills in k_scan_h16 cost 10-15 % of the headline throughput)."""
    import re
    from pathlib import Path

    import lynsedb_amd  # noqa: F401  (build() has produced the library and the resource report)

    rep = Path(lynsedb_amd.__file__).parent / "csrc" / "resource_usage.txt"
    if not rep.exists():
        pytest.skip("resource report not produced by this build")
    text = rep.read_text()
    blocks = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", text, flags=re.S)
    assert blocks, "no resource remarks in the report"
    seen = 0
    # k_scan_h16<WQ, WR, TQ, TR, METRIC, NSV, NSQ, NT, TILED, RAG, DBG, FILT, I8Q, EMIT>: every unfiltered instantiation
    # of the product build (all emission modes, ragged or not, f16 shadow / SQ8 / certified int8) must be spill-free;
    # the subset-filter variants of the <4,2,2,4> tiling may keep a small epilogue spill (bounded here).
    pat = re.compile(r"k_scan_h16ILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi\dELi\dELi\dELb([01])ELb([01])ELi(\d+)ELb([01])ELi(\d)ELi(n?\d)E")
    hot = {"ip_f16": 0, "ip_i8c": 0, "l2": 0, "cos": 0, "small": 0, "fused_sample": 0}
    for name, scratch in blocks:
        m = pat.search(name)
        if m:
            wq, wr, tq, tr, metric, tiled, rag, dbg, filt, i8q, emit = m.groups()
            if dbg != "0":
                continue
            seen += 1
            fused_sample = re.search(r"ELi1EEEvNS_8ScanArgsE$", name) is not None   # ... FS = 1>: the fused sample stage
            if filt == "1" and (wq, wr) == ("4", "2"):
                assert int(scratch) <= 128, (name, scratch)
            elif fused_sample:
                # one value parked in scratch in the prologue and reloaded in the once-per-launch threshold hand-over (the
                # generated code has no scratch access between the slab barrier and the last MFMA of the loop: `make asm`)
                assert int(scratch) <= 32, (name, scratch)
                hot["fused_sample"] = hot.get("fused_sample", 0) + 1
            else:
                assert int(scratch) == 0, (name, scratch)
            if (wq, wr, filt, rag, emit, tiled) == ("2", "4", "0", "0", "0", "0") and metric == "0":
                hot["ip_i8c" if i8q == "2" else "ip_f16"] += 1
            if (wq, wr, filt, rag, emit, tiled) == ("4", "2", "0", "0", "0", "0") and i8q == "0":
                hot["l2" if metric == "1" else "cos"] += 1
            if (wq, wr, filt, rag, tiled) == ("1", "4", "0", "0", "0"):
                hot["small"] += 1
        if "k_scan_binary_rows" in name and "ILi0ELi16ELb0" in name:
            seen += 1
            assert int(scratch) == 0, (name, scratch)
    assert all(v >= 1 for v in hot.values()), hot   # the kernels bench.py / the BASELINE configs run were all seen
    assert seen >= 20


def test_reference_partition_rule_matches_the_golden_hashes(golden_dir):
    """blake2b64("{db}/{coll}/{id}") % 4096 -> bucket % world (python/lynse/cluster.py:156-158, :1273, :1364-1370): the
    values were captured from the reference's own `_hash_u64` (tests/golden/make_python_reference_vectors.py)."""
    import json

    from lynsedb_amd.sharded import bucket_of_id, hash_u64, shard_of_id

    g = json.loads((golden_dir / "python_reference_vectors.json").read_text())["hash_u64"]
    assert len(g) >= 4
    for e in g:
        assert hash_u64(e["key"]) == int(e["hash"]), e
        assert hash_u64(e["key"]) % 4096 == e["bucket4096"]
        if e["key"].count("/") != 2:
            continue
        db, coll, item = e["key"].split("/")
        assert bucket_of_id(db, coll, item) == e["bucket4096"]
        for world in (1, 2, 4, 8):
            assert shard_of_id(db, coll, item, world) == e["bucket4096"] % world


def test_lynse_hip_devices_env_selects_the_device_of_a_rank(monkeypatch):
    """`LYNSE_HIP_DEVICES`: the ordinals this process may use; LOCAL_RANK indexes the list, LYNSE_HIP_DEVICE overrides."""
    import lynsedb_amd as L

    for var in ("LYNSE_HIP_DEVICE", "LOCAL_RANK", "LYNSE_HIP_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert L.default_device() == 0
    monkeypatch.setenv("LYNSE_HIP_DEVICES", "2, 3,5")
    assert L.visible_devices() == [2, 3, 5] and L.default_device() == 2
    monkeypatch.setenv("LOCAL_RANK", "4")
    assert L.default_device() == 3                      # entry 4 % 3
    monkeypatch.setenv("LYNSE_HIP_DEVICE", "7")
    assert L.default_device() == 7
    monkeypatch.delenv("LYNSE_HIP_DEVICE")
    monkeypatch.setenv("LYNSE_HIP_DEVICES", "x,1")
    with pytest.raises(ValueError):
        L.default_device()
    monkeypatch.delenv("LYNSE_HIP_DEVICES")
    assert L.default_device() == 4                      # LOCAL_RANK alone: one process per GPU
