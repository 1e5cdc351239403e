#!/bin/bash
set -u
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for cfg in "MOEINF_SH2_NW=4" "MOEINF_SH2_NW=8" "MOEINF_SH2_NW=16" "MOEINF_SH2_NW=16 MOEINF_SH2_U=8" "MOEINF_SH2_NW=8 MOEINF_SH2_U=8" "MOEINF_SH2_NW=8 MOEINF_SH1_U=8" "MOEINF_SH2_NW=16 MOEINF_SH1_U=8"; do
  echo "== $cfg"; env $cfg python tools/ds_experiments.py base 2>&1 | tail -1
done
