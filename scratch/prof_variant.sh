#!/bin/bash
# prof_variant.sh <variant> <tag> <program> [args...]: prof_cmd.sh with clustering_amd/lib/variants/<variant>.so as the library
cd $GRAFT_REPO_ROOT
cp clustering_amd/lib/libdcdensity.so /tmp/lib_saved.so
# (whatever ends this script -- a timeout, a kill of prof_cmd.sh -- the product library comes back: ADVICE r4)
trap 'cp /tmp/lib_saved.so $GRAFT_REPO_ROOT/clustering_amd/lib/libdcdensity.so' EXIT
cp clustering_amd/lib/variants/$1.so clustering_amd/lib/libdcdensity.so
shift
bash scratch/prof_cmd.sh "$@" > /dev/null
cd $GRAFT_REPO_ROOT
cp /tmp/lib_saved.so clustering_amd/lib/libdcdensity.so
python3 - "$1" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}_pmc.json"))
for k,v in d.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES",0) < 1e9: continue
    ch=v["SQ_VALU_MFMA_BUSY_CYCLES"]/(6*32)
    print(k[:60], "chains %.3g"%ch, "VALU/chain %.1f"%(v["SQ_INSTS_VALU"]/ch), "SALU/chain %.1f"%(v["SQ_INSTS_SALU"]/ch), "LDS/chain %.1f"%(v["SQ_INSTS_LDS"]/ch),
          "wave_cyc/chain %.0f"%(v["SQ_WAVE_CYCLES"]/ch), "wait_any %.2f"%(v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"]), "wait_inst %.2f"%(v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"]),
          "active_valu %.2f"%(v["SQ_ACTIVE_INST_VALU"]/v["SQ_WAVE_CYCLES"]), "busy_cyc %.3g"%v["SQ_BUSY_CYCLES"], "gui %.3g"%v["GRBM_GUI_ACTIVE"], "waves %d"%v["SQ_WAVES"],
          "rd %.3g wr %.3g"%(v["TCC_EA0_RDREQ_sum"],v["TCC_EA0_WRREQ_sum"]), "lds_active %.3g bank_conf %.3g"%(v["SQ_ACTIVE_INST_LDS"], v.get("SQ_LDS_BANK_CONFLICT",0)))
PY
