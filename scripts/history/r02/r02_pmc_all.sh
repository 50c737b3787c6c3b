for lib in r1 v1; do
bash scripts/history/r02/r02_pmc.sh $lib c2 -- 
bash scripts/history/r02/r02_pmc.sh $lib c1 -- --num-samples 4000 --num-ants 1 --blocks 16384
bash scripts/history/r02/r02_pmc.sh $lib c3 GAT_MC_MODE=0 GAT_DC_KT=1 -- --baseline-config 2
done
