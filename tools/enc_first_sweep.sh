# developer sweep: first-poll delay of the cooperative encoder BiLSTM (x 512 clocks): encoder ms of the headline utterance
for f in 0 1 2 3 4; do echo -n "XDTTS_ENC_FIRST=$f: "; XDTTS_ENC_FIRST=$f timeout 120 python tools/headline_call.py 6 2>&1 | grep "^call [345]" | sed "s/.*'encoder_ms': \([0-9.]*\).*/\1/" | tr '\n' ' '; echo; done
