import re,sys,subprocess
for f in sys.argv[1:]:
    txt=open(f).read()
    i=txt.find('amdhsa.kernels:')
    for blk in txt[i:].split('- .agpr_count')[1:]:
        g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
        name=subprocess.run(['c++filt',g('name')],capture_output=True,text=True).stdout.strip()
        name=re.sub(r'\(hipnmf::\w+<\w+>\)','',name).replace('hipnmf::','').replace('void ','')
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0].strip(':'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} vspill {g('vgpr_spill_count'):>4} sspill {g('sgpr_spill_count'):>4} lds {g('group_segment_fixed_size')}")
