set -e
python3 tools/rep_ab.py 1000000 bench "" PLL_AMD_REP_WGS=16 PLL_AMD_REP_WGS=32 ""  PLL_AMD_REP_WGS=16
python3 tools/rep_ab.py 125000 bench "" PLL_AMD_REP_WGS=16 PLL_AMD_REP_WGS=4 ""
