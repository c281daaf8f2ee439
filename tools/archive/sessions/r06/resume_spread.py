"""run-to-run spread of the train entry's epoch losses (the resume test's tolerance)"""
import io, os, sys, contextlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))))
from pseldnets_amd import train
for lr in ('0.001', '0.0001'):
    argv = ['experiment=synth_maccdoa', 'model.kwargs.embed_dim=48', 'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]',
            'model.batch_size=4', 'data.num_classes=5', 'trainer.limit_train_batches=5', f'model.optimizer.kwargs.lr={lr}', 'augment=default']
    for rep in range(5):
        d = tempfile.mkdtemp()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            train.main(argv + ['trainer.max_epochs=2', f'paths.output_dir={d}'])
        print(lr, rep, [ln.split('  lr')[0].split('loss_all')[1].strip() for ln in buf.getvalue().splitlines() if ln.startswith('epoch')])
