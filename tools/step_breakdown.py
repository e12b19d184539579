"""Serialised (device-synchronised) timing of every part of a supernet pretrain step and of a search iteration: captured
passes, eager passes (forward / backward), latency model, optimizers.  Run on an MI355X:  python tools/step_breakdown.py"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd.train_step import SupernetStep
from fasterseg_amd import kernels as K
from fasterseg_amd import latency_lookup_table
lut = latency_lookup_table.load_shipped('bf16')
DT = torch.bfloat16 if os.environ.get('FS_BREAKDOWN_DTYPE', 'bf16') == 'bf16' else torch.float32
ONLY = os.environ.get('FS_BREAKDOWN_ONLY', '')

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

def make(b, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(b, 3, h, w, generator=g).cuda(), torch.randint(0, 19, (b, h // 8, w // 8), generator=g).cuda()

for pretrain in (True, False):
    if ONLY and ONLY != ('pretrain' if pretrain else 'search'):
        continue
    st = SupernetStep(pretrain=pretrain, lut=None if pretrain else lut, compute_dtype=DT)
    b, h, w = (3, 256, 512) if pretrain else (2, 224, 448)
    imgs, tgt = make(b, h, w, 1); imgs_s, tgt_s = make(b, h, w, 2)
    from fasterseg_amd import functional as FN
    FN.set_compute_dtype(DT)
    for _ in range(2):
        st.step(imgs, tgt, imgs_s, tgt_s)
    FN.set_compute_dtype(DT)
    T = {}
    def timed(name, fn):
        t0 = sync(); r = fn(); T[name] = T.get(name, 0) + (sync() - t0) * 1e3; return r
    def phase(ph, im, tg):
        s_i, s_t = st.static[ph]; s_i.copy_(im); s_t.copy_(tg)
        for group in st._pass_groups:
            if st._is_static(group[0]):
                g, loss, touched = st.graphs[(ph, group)]
                for spec in group: st._select(spec)
                timed("%s graph %s" % (ph, group), g.replay)
                if touched is not None: st.sync.mark_touched(touched)
            else:
                loss = timed("%s eager fwd %s" % (ph, group), lambda: st._run_group(group, im, tg))
                timed("%s eager bwd %s" % (ph, group), loss.backward)
    t_all = sync()
    if not pretrain:
        st._set_phase("a"); st._zero_arch_grads(); K.zero_pool.reset(imgs.device)
        phase("a", imgs_s, tgt_s)
        ll = timed("latency fwd", st.architect._latency_loss)
        timed("latency bwd", ll.backward)
        K.zero_pool.stop()
        timed("arch optim", lambda: [o.step() for o in st.architect.optimizers])
    st._set_phase("w")
    timed("prepare", st.sync.prepare)
    phase("w", imgs, tgt)
    timed("sync", st.sync.sync)
    timed("clip+sgd", st.optimizer.step)
    total = (sync() - t_all) * 1e3
    print("==== pretrain" if pretrain else "==== search", "serialised total %.1f ms" % total)
    for k, v in T.items():
        print("  %-34s %8.1f ms" % (k, v))
    del st
    torch.cuda.empty_cache()
