import sys; sys.path.insert(0, "/root/repo")
import bench
from clsr_amd.ops import query
from clsr_amd.net import CLSRNet
w = bench.Workload("taobao")
n = w.net
hp = n.hp
print(dict(sw=n.enc_bwd_fused, bf16=n.bf16, typ=type(n) is CLSRNet, defer=n.defer_dw, kind=n._t4_kind, ie=hp.interest_evolve,
           ma=hp.manual_alpha, pls=hp.predict_long_short, enc_in=n.enc_in, D=n.D, Du=n.Du, H=n.H,
           sup=query("clsr_enc_bwd_fused_supported", n.D, n.H, 480), g1=n._enc_off("g1"), t4=n._enc_off("t4"), g2=n._enc_off("g2"),
           chunks=n.rnn_chunks))
