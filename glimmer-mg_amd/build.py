"""Build lib/libgmg.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import fcntl
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(PKG, "lib", "libgmg.so")
SOURCES = ["csrc/gmg_api.hip", "csrc/gmg_kernels.hip", "csrc/gmg_frame6.hip", "csrc/gmg_orfs.hip", "csrc/gmg_mg.hip", "csrc/gmg_ingest.hip", "csrc/gmg_strings.hip", "csrc/gmg_train.hip", "host/icm.cc", "host/icm_train.cc", "host/gmg_icm_c.cc", "host/gmg_shard.cc", "host/gmg_classes.cc"]
HEADERS = ["csrc/gmg_internal.h", "csrc/gmg_device.h", "csrc/gmg_scan.h", "csrc/gmg_mg_errtile.h", "csrc/gmg_mg_errwave.h", "csrc/gmg_mg_orfbits.h", "host/icm.hh", "../include/gmg.h", "../include/gmg_icm.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(PKG, f)) > t for f in SOURCES + HEADERS)


def build_variant(name, defines):
    """lib/variants/libgmg_<name>.so: the same sources with extra -D switches (kernel A/B runs; GMG_LIB_PATH selects one)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(PKG, "lib", "variants", "libgmg_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc, *FLAGS, *["-D" + d for d in defines], "-o", out, *SOURCES]
    res = subprocess.run(cmd, cwd=PKG, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        print(res.stdout)
        raise RuntimeError("hipcc failed building " + out)
    return out


def build_lib(force=False, verbose=False):
    """Compile every HIP/C++ source into lib/libgmg.so.  Returns the library path."""
    if os.environ.get("GMG_LIB_PATH"):
        return os.environ["GMG_LIB_PATH"]
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # several ranks of one job may get here at once (bench.py under torch.distributed.run): one of them builds, into a
    # temporary file that replaces the library atomically; the others wait for the lock and find the library fresh
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return LIB
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            # one object per source, only the stale ones, a few at a time (the box has 8 - 16 cores); then one link
            objdir = os.path.join(PKG, "lib", "obj")
            os.makedirs(objdir, exist_ok=True)
            # the objects belong to ONE toolchain and ONE set of flags: a stamp beside them (flags, this file, hipcc --version);
            # anything else in there is rebuilt
            import hashlib
            ver = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
            stamp = hashlib.sha256((" ".join(FLAGS) + open(os.path.abspath(__file__)).read() + ver).encode()).hexdigest()
            stamp_file = os.path.join(objdir, "STAMP")
            if not os.path.exists(stamp_file) or open(stamp_file).read() != stamp:
                force = True
            newest_header = max(os.path.getmtime(os.path.join(PKG, h)) for h in HEADERS)
            objs, jobs = [], []
            for src in SOURCES:
                obj = os.path.join(objdir, os.path.basename(src) + ".o")
                objs.append(obj)
                if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(os.path.join(PKG, src)), newest_header):
                    jobs.append((src, obj, [hipcc, *[f for f in FLAGS if f != "-shared"], "-c", "-o", obj + ".tmp", src]))
            running = []
            failed = []

            def reap(block):
                for item in list(running):
                    src, obj, proc = item
                    if block or proc.poll() is not None:
                        out = proc.communicate()[0]
                        running.remove(item)
                        if verbose or proc.returncode != 0:
                            print(out)
                        if proc.returncode != 0:
                            failed.append(src)
                        else:
                            os.replace(obj + ".tmp", obj)           # (an interrupted compile leaves no half-written object behind)
                        if block:
                            return

            width = max(1, min(6, (os.cpu_count() or 2) - 1))
            for src, obj, cmd in jobs:
                while len(running) >= width:
                    reap(True)
                running.append((src, obj, subprocess.Popen(cmd, cwd=PKG, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            while running:
                reap(True)
            if failed:
                raise RuntimeError("hipcc failed compiling " + ", ".join(failed))
            with open(stamp_file, "w") as f:
                f.write(stamp)
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
            res = subprocess.run(cmd, cwd=PKG, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if verbose or res.returncode != 0:
                print(" ".join(cmd))
                print(res.stdout)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed linking libgmg.so")
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":                              # python3 glimmer-mg_amd/build.py [--force]
    import sys
    print(build_lib(force="--force" in sys.argv[1:], verbose="-v" in sys.argv[1:]))
