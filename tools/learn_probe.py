"""Learning sanity run (from stylegan-for-facerec_amd/: PYTHONPATH=. FRHIP_COMPUTE_DTYPE=bf16 python ../tools/learn_probe.py):
train.py on 50 synthetic identities x 20 images, pSp IR-SE-50, bf16, 12 epochs.  Round 1 on one MI355X: training loss 37.7 ->
0.000, Prec@1 0 -> 100 %; with GPU_INPUT_PIPELINE=True, SHARDED_HEAD=True (structured staged identities under random
resize-crop-flip) Prec@1 reaches 97 %."""
import sys, runpy
import configs.config_synthetic_smoke as c
c.configurations[1].update(BATCH_SIZE=100, NUM_EPOCH=12, MODEL_ROOT='/tmp/learn_model', LOG_ROOT='/tmp/learn_log', LR=0.03, GPU_INPUT_PIPELINE=False)
sys.argv = ['train.py', '--config', 'configs/config_synthetic_smoke.py', '--synthetic', '50x20']
runpy.run_path('train.py', run_name='__main__')
