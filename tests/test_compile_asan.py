"""The host engine under AddressSanitizer / UBSan on random projects, without a GPU (tests/asan_compile.cpp, tests/mock_hip.cpp).

csrc/engine.cpp's event compiler and descriptor / arena builder (compile_chunk, submit_chunk: byte offsets into staging
arenas, pointer fix-ups, scratch and hand-off regions sized on the host and filled by kernels) only ever ran compiled by
hipcc into the GPU library.  Here the SAME sources -- engine.cpp, project.cpp, the Lua subset, WAV and MIDI readers --
build with g++ -fsanitize=address,undefined against a host-memory stand-in for the HIP runtime whose "launches" walk their
descriptor tables and both ends of every array a descriptor points to; tests/test_gpu_fuzz.py's random_project generator
(sine / synth kinds and band-pass chains included) is pushed through the real front-end and C ABI in all three band modes,
un-chunked and in 4 096-frame chunks (band_mode 2 with every audited render done again), fresh / scanned / continued renders
and block pulls.  CPU only; TD_ASAN_SEEDS=<n> widens the run (profiles/r05_compile_asan.txt: 5 000 seeds)."""
import multiprocessing
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "termdaw_amd", "csrc")
SOURCES = ["engine.cpp", "compile.cpp", "devmem.cpp", "comm.cpp", "project.cpp", "lua_subset.cpp", "wav.cpp", "midi.cpp"]


def _build(out_dir):
    flags = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
             "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", CSRC, "-I", os.path.join(ROOT, "include")]
    jobs = [(os.path.join(CSRC, f), os.path.join(out_dir, f + ".o")) for f in SOURCES]
    jobs += [(os.path.join(ROOT, "tests", f), os.path.join(out_dir, f + ".o")) for f in ("mock_hip.cpp", "asan_compile.cpp")]
    procs = [subprocess.Popen(["g++"] + flags + ["-c", src, "-o", obj]) for src, obj in jobs]
    for p in procs:
        assert p.wait() == 0
    exe = os.path.join(out_dir, "asan_compile")
    subprocess.check_call(["g++", "-fsanitize=address,undefined", "-o", exe] + [o for _, o in jobs] + ["-lpthread", "-ldl"])
    return exe


def _write_projects(args):
    base, seeds = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fuzz as F
    dirs = []
    for seed in seeds:
        p = F.random_project(seed, allow_sinf=True)
        d = os.path.join(base, "s%d" % seed)
        lua = p.to_lua(os.path.join(d, "assets"))
        with open(os.path.join(d, "project.lua"), "w") as f:
            f.write(lua)
        with open(os.path.join(d, "meta.txt"), "w") as f:
            f.write(str(p.bl))
        dirs.append(d)
    return dirs


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="needs g++ and the HIP headers")
def test_event_compiler_and_descriptor_builder_under_sanitizers(tmp_path):
    exe = _build(str(tmp_path))
    n = int(os.environ.get("TD_ASAN_SEEDS", "600"))
    workers = max(1, min(8, os.cpu_count() or 1))
    # (TD_ALLOC_CACHE_MB=0: every "device" block is its own malloc of exactly the size asked for, so that one byte past it is a report)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TD_ALLOC_CACHE_MB="0")
    renders = 0
    with multiprocessing.Pool(workers) as pool:
        for lo in range(0, n, 200):      # (a few hundred projects on disk at a time)
            seeds = list(range(lo, min(n, lo + 200)))
            base = str(tmp_path / ("p%d" % lo))
            chunks = [seeds[i::workers] for i in range(workers)]
            dir_lists = pool.map(_write_projects, [(base, c) for c in chunks if c])
            procs = [subprocess.Popen([exe] + dl, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for dl in dir_lists]
            if lo == 0:   # ... and one worker's projects again with the memory cache ON and a budget it overruns (csrc/devmem.cpp:
                # blocks handed out again, trimmed largest first -- its bookkeeping under the sanitizers)
                procs.append(subprocess.Popen([exe] + dir_lists[0], env=dict(env, TD_ALLOC_CACHE_MB="16"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
            for p in procs:
                out, err = p.communicate(timeout=1200)
                assert p.returncode == 0, (out[-500:], err[-4000:])
                assert "asan_compile done" in out
                renders += int(out.split(" renders")[0].split()[-1])
            shutil.rmtree(base, ignore_errors=True)
    assert renders >= n * 17          # (3 band modes x 2 chunkings x 3 renders per accepted project)
    print("asan_compile: %d projects, %d renders clean" % (n, renders))
