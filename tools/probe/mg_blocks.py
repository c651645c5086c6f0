"""What the blocks of the 256^3 hierarchy look like (chunks, encodings, bytes), per level."""
import os, sys
root, tag = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sys.path.insert(0, root)
from __graft_entry__ import load_package
pa = load_package()
S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
for lev in range(S.l - 1, 0, -1):
    blk = S.A_vec[lev].matrix_partition.items[0].own_own
    print(f"[{tag}] level {lev} A: {blk.info()} enc {blk.encoding()} xwin {blk.xwin()} dev {blk.device_bytes()} stream {blk.stream_bytes()}")
    for k, cb in enumerate(S.gs_states[lev].parts.items[0][0][:2]):
        print(f"[{tag}] level {lev} colour {k}: {cb.info()} enc {cb.encoding()} xwin {cb.xwin()} dev {cb.device_bytes()} stream {cb.stream_bytes()}")
    if S.row_blocks and S.row_blocks[lev - 1] is not None:
        rb = S.row_blocks[lev - 1].items[0]
        print(f"[{tag}] level {lev} row block: {rb.info()} enc {rb.encoding()} xwin {rb.xwin()} stream {rb.stream_bytes()}")
