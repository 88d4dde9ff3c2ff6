# numpy emulation of the in-place Stockham radix schedule used by amt_stft.hip
import numpy as np
def dft_small(x):  # x: [R, ...]
    R=x.shape[0]
    W=np.exp(-2j*np.pi*np.outer(np.arange(R),np.arange(R))/R)
    return np.tensordot(W,x,axes=(1,0))
def stockham(z, radices):
    N=len(z); buf=z.astype(np.complex128).copy(); Ns=1
    for R in radices:
        nb=N//R
        j=np.arange(nb)
        k=j%Ns
        inp=np.stack([buf[j+r*nb] for r in range(R)])           # reads
        tw=np.exp(-2j*np.pi*np.outer(np.arange(R),k)/(Ns*R))
        # table form: W[m], m = r*k*(N/(Ns*R))
        m=(np.outer(np.arange(R),k)*(N//(Ns*R)))
        assert m.max()<N
        tw2=np.exp(-2j*np.pi*m/N); assert np.allclose(tw,tw2)
        y=dft_small(inp*tw)
        out=np.empty_like(buf)
        base=(j-k)*R+k
        for r in range(R): out[base+r*Ns]=y[r]
        buf=out; Ns*=R
    return buf
for N,rad in ((2048,(8,8,8,4)),(4096,(8,8,8,8)),(1024,(8,8,4,4)),(512,(8,8,8)),(256,(8,8,4))):
    z=np.random.randn(N)+1j*np.random.randn(N)
    print(N, np.abs(stockham(z,rad)-np.fft.fft(z)).max())
# two-real-frames trick
N=2048
a=np.random.randn(N); b=np.random.randn(N)
Z=np.fft.fft(a+1j*b); k=np.arange(N//2+1); Zm=np.conj(Z[(N-k)%N])
X1=(Z[k]+Zm)/2; X2=(Z[k]-Zm)/(2j)
print(np.abs(X1-np.fft.rfft(a)).max(), np.abs(X2-np.fft.rfft(b)).max())
# inverse: Z[k]=X1+iX2 ; Z[N-k]=conj(X1)+i conj(X2)
Zr=np.empty(N,complex); Zr[k]=X1+1j*X2; kk=np.arange(1,N//2); Zr[N-kk]=np.conj(X1[kk])+1j*np.conj(X2[kk])
z=np.fft.ifft(Zr); print(np.abs(z.real-a).max(), np.abs(z.imag-b).max())
