"""Build record of the in-tree libraries: toolchain, flags, kernel-source hash and the sha256 of every .so `make -C mlx-video_amd/csrc`
produced (the .so files are git-ignored; they travel to the GPU box with the snapshot).  python scripts/build_record.py OUT.json"""
import hashlib, json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import source_sha
pkg = os.path.join(root, "mlx-video_amd")
mk = open(os.path.join(pkg, "csrc", "Makefile")).read()
rec = {"hipcc": subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.strip().splitlines()[:3],
       "arch": re.search(r"ARCH \?= (\S+)", mk).group(1), "cxxflags": re.search(r"CXXFLAGS = (.*)", mk).group(1),
       "per_file": "attention.hip: -fno-slp-vectorize; libltxk_ab.so: the same sources with -DLTXK_AB",
       "kernel_source_sha": source_sha(), "git_head": subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip(),
       "libraries": {f: hashlib.sha256(open(os.path.join(pkg, f), "rb").read()).hexdigest() for f in sorted(os.listdir(pkg)) if f.endswith(".so")},
       "recipe": "make -C mlx-video_amd/csrc  (python -c 'import __graft_entry__ as g; g.build()')"}
json.dump(rec, open(sys.argv[1], "w"), indent=1)
print(json.dumps(rec, indent=1))
