import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import torch, las_oracle as lo
from ss_asr_amd import ops
torch.manual_seed(0)
for (B,T,lens) in [(2,375,[375,200]),(2,130,[130,90]),(2,128,[128,100]),(2,256,[256,256]),(2,257,[257,100])]:
    A,E,D=128,512,256
    feat=torch.randn(B,T,E); comp=torch.tanh(torch.randn(B,T,A)); state=torch.randn(B,D); w=torch.randn(A,D)/16
    al,cx=lo.attention_step_explicit(state.double(),feat.double(),comp.double(),lens,w.double())
    ld=torch.tensor(lens,dtype=torch.int32,device='cuda')
    ad,cd=ops.attn_step(state.cuda(),w.cuda(),comp.cuda(),feat.cuda(),ld)
    err=(ad.cpu().double()-al).abs()
    bad=(err>1e-5).nonzero()
    print(T,lens,'max err',err.max().item(),'nbad',len(bad), bad[:6].tolist(), 'ctx err',(cd.cpu().double()-cx).abs().max().item())
