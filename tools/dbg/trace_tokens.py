"""Pure-Python RFC 1951 token tracer (debug aid): lists tokens around an output offset."""
import sys
sys.path.insert(0, '/root/repo')
import synth

LBASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
ORDER=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]

class BR:
    def __init__(s,d): s.d=d; s.p=0
    def bits(s,n):
        v=0
        for i in range(n):
            v |= ((s.d[s.p>>3]>>(s.p&7))&1)<<i; s.p+=1
        return v
def mk(lens):
    codes={}; code=0
    cnt=[0]*16
    for l in lens: cnt[l]+=1
    cnt[0]=0; nxt=[0]*16
    for i in range(1,16):
        code=(code+cnt[i-1])<<1; nxt[i]=code
    for s,l in enumerate(lens):
        if l: codes[(l,nxt[l])]=s; nxt[l]+=1
    return codes
def dec(br,codes):
    c=0
    for l in range(1,16):
        c=(c<<1)|br.bits(1)
        if (l,c) in codes: return codes[(l,c)], l
    raise ValueError
def trace(raw, lo, hi):
    br=BR(raw); op=0; out=[]
    while True:
        fin=br.bits(1); t=br.bits(2)
        print("block type",t,"at out",op,"bitpos",br.p)
        if t==0:
            br.p=(br.p+7)&~7; ln=br.bits(16); br.bits(16); br.p+=8*ln; op+=ln
        else:
            if t==1:
                ll=[8]*144+[9]*112+[7]*24+[8]*8; dl=[5]*30
            else:
                hl=br.bits(5)+257; hd=br.bits(5)+1; hc=br.bits(4)+4
                cl=[0]*19
                for i in range(hc): cl[ORDER[i]]=br.bits(3)
                cc=mk(cl); lens=[]
                while len(lens)<hl+hd:
                    s,_=dec(br,cc)
                    if s<16: lens.append(s)
                    elif s==16: lens+= [lens[-1]]*(br.bits(2)+3)
                    elif s==17: lens+=[0]*(br.bits(3)+3)
                    else: lens+=[0]*(br.bits(7)+11)
                ll=lens[:hl]; dl=lens[hl:]
            lc=mk(ll); dc=mk(dl)
            while True:
                bp0=br.p
                s,sl=dec(br,lc)
                if s<256:
                    if lo<=op<hi: print(" lit",op,s,"codelen",sl)
                    op+=1
                elif s==256: break
                else:
                    L=LBASE[s-257]+br.bits(LEXT[s-257])
                    d,dl_=dec(br,dc); D=DBASE[d]+br.bits(DEXT[d])
                    if op+L>lo and op<hi: print(" match at",op,"len",L,"dist",D,"lencode",s,"lcodelen",sl,"dcode",d,"dcodelen",dl_,"bits",br.p-bp0, "bitpos", bp0)
                    op+=L
        if fin: break
name,lo=sys.argv[1],int(sys.argv[2])
raw=synth.fixture(name)[10:-8]
trace(raw,lo-40,lo+140)
