#!/bin/bash
# SQ instruction counters of match_kernel with parts of it disabled (library variant built with -DLZSS_DEBUG_KNOBS;
# MI355_LZSS_ABLATE bits: 1 dictionary, 2 take SA[kdn] for the smallest-rank occurrence, 4 walks, 16 cooperative tails)
cp compress_amd/variants/lib_dbg.so compress_amd/libmi355_lzss.so || exit 1
for a in ${ABL:-0 1 4 16 5}; do
  echo "ablate=$a"
  MI355_LZSS_ABLATE=$a bash $GRAFT_REPO_ROOT/scripts/pmc.sh abl$a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" 2>&1 | grep -E "match_kernel|no csv" || exit 1
done
