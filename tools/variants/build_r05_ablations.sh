#!/bin/bash
# The back-end ablations of round 5 (price the non-term part of k_idct_color<1>).  Results wrong, timing valid.
# Every variant is the pair-form kernel WITHOUT its term loop (JS_EXP_NOTERMS) minus one more part; run with tools/ab_round.sh.
set -e
cd $(dirname $0)/../..
P="-p tools/variants/ablations.patch"
tools/build_variant.sh a_noterms            $P -DJS_EXP_NOTERMS
tools/build_variant.sh b_nocolor            $P -DJS_EXP_NOTERMS -DJS_EXP_NOCOLOR
tools/build_variant.sh c_nodibstore         $P -DJS_EXP_NOTERMS -DJS_EXP_NODIBSTORE
tools/build_variant.sh d_nocolormath        $P -DJS_EXP_NOTERMS -DJS_EXP_NOCOLORMATH
tools/build_variant.sh e_noloads            $P -DJS_EXP_NOTERMS -DJS_EXP_NOLOADS
tools/build_variant.sh f_notile             $P -DJS_EXP_NOTERMS -DJS_EXP_NOTILE
tools/build_variant.sh g_nobright           $P -DJS_EXP_NOTERMS -DJS_EXP_NOBRIGHT
tools/build_variant.sh h_listonly           $P -DJS_EXP_LISTONLY
tools/build_variant.sh i_bare               $P -DJS_EXP_NOTERMS -DJS_EXP_NOCOLOR -DJS_EXP_NOLOADS -DJS_EXP_NOTILE
