import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import test_hip_parity_r3 as t
name='ca_b64_K1000'
cfg, sd, pb, K, window, noise = t.g14_case(name)
B=len(pb.size)
G=t.G14
margins=G[name+'/margins']; steps=G[name+'/ckpt_steps']; ck_z=G[name+'/ckpt_z']
want=G[name+'/xh_phar']
for engine in ['split','fp32']:
    h=t.new_handle(cfg,sd); h.set_gemm_mode(engine=='split'); h.set_layout(pb.num_nodes_phar,pb.size); h.set_step_table(K,t.host_step_table(cfg,K))
    got,got_p,z_steps=h.sample_chain(t.dev(pb.x),t.dev(pb.one_hot),K,noise=t.dev(noise.numpy()),want_steps=True,use_graph=False)
    cum=np.minimum.accumulate(margins,axis=0)
    for i,s in enumerate(steps):
        z=z_steps[int(s)-1].cpu().numpy()
        e=t.per_sample_rms(z[:,:3],ck_z[i][:,:3],B)
        o=np.argsort(-e)[:6]
        print(engine,int(s),'median %.2e'%np.median(e),'n<=1e-4:',int((e<=1e-4).sum()),'worst:',[(int(b),'%.1e'%e[b],'%.1e'%cum[(int(s)-1)//window][b]) for b in o])
    e=t.per_sample_rms(got.cpu().numpy()[:,:3],want[:,:3],B)
    o=np.argsort(-e)[:8]
    print(engine,'final median %.2e'%np.median(e),'n<=1e-4:',int((e<=1e-4).sum()),[(int(b),'%.1e'%e[b],'%.1e'%cum[-1][b]) for b in o])
    h.close()
