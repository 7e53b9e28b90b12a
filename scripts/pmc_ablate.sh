#!/bin/bash
# SQ instruction counters of match_kernel with parts of it disabled (MI355_LZSS_ABLATE bits: 1 dict, 2 leftmost, 4 walks)
for a in 0 1 2 4 7; do
  echo "ablate=$a"
  MI355_LZSS_ABLATE=$a bash $GRAFT_REPO_ROOT/scripts/pmc.sh abl$a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" 2>&1 | grep -E "match_kernel|no csv" || exit 1
done
