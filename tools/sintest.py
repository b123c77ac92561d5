import torch
x=torch.rand(1000000)*6434
g=torch.sin(x.cuda()).cpu().double(); c=torch.sin(x).double(); t=torch.sin(x.double())
print('gpu sin err',(g-t).abs().max().item(),'cpu sin err',(c-t).abs().max().item())
