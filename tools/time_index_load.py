"""Times the index upload paths on the GPU box: bmf_load_index_files (streamed .qgram) twice, a plain Python read of
the same file, and bmf_load_index from memory.  Expects the files tools/e2e_cli.py leaves in /tmp/bm_e2e.

    python tools/e2e_cli.py --reads 1000 && python tools/time_index_load.py
"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bucket-map_amd", "python"))
import numpy as np
import bucket_map_amd as bma
d = "/tmp/bm_e2e"
nb = 26413
t = time.perf_counter(); f = bma.Filter(bma.Params.from_cli(nb, read_len=300)); print("create", round(time.perf_counter() - t, 3))
t = time.perf_counter(); f.load_index_files(d, "idx"); print("load 1", round(time.perf_counter() - t, 3))
f.reset()
t = time.perf_counter(); f.load_index_files(d, "idx"); print("load 2", round(time.perf_counter() - t, 3))
f.reset()
t = time.perf_counter(); raw = open(d + "/idx.qgram", "rb").read(); print("python read", round(time.perf_counter() - t, 3), len(raw))
t = time.perf_counter(); k2i = np.loadtxt(d + "/idx.kmers_index", dtype=np.int32); print("loadtxt", round(time.perf_counter() - t, 3))
rows = np.frombuffer(raw, np.uint8)
t = time.perf_counter(); f.load_index(rows, k2i); print("load from memory", round(time.perf_counter() - t, 3))
