"""The host-only surface of the C ABI (csrc/wav_io.cpp: RIFF parsing of untrusted files, batched decode, file copies; csrc/errors.cpp) under
AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers).  ``make -C adt_str_amd/csrc asan`` builds the same sources with
``-fsanitize=address,undefined`` into ``libadt_host_asan.so``; tests/test_host_io_cpu.py (200 random RIFF layouts, truncated / lying headers,
NaNs, copies) then runs against it in a child interpreter with the ASan runtime preloaded.  Device code cannot be sanitized on this pool
(no GPU ASan), so this covers exactly the code that reads bytes it did not produce."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_io_suite_is_clean_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, "adt_str_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "adt_str_amd", "libadt_host_asan.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.exists(lib) and os.path.isabs(asan_rt) and os.path.exists(asan_rt)
    env = dict(os.environ, ADT_HOST_ONLY_LIB=lib, LD_PRELOAD=asan_rt, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_io_cpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-6000:]
    assert r.returncode == 0, out[-6000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-2000:]
