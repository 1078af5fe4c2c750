"""utils/visualizer.py of the reference, text part only (the visdom plots are outside the hot path): the per-iteration
log line and the log file."""
import os


class Visualizer(object):
    def __init__(self, opt=None, log_path=None):
        self.opt = opt
        self.log_path = log_path

    def print_log(self, epoch, iter, total_iter, value1, value2=None):
        """utils/visualizer.py:66-80: 'epoch:E, iters:I/N k: v.vvvv ...' (+ a rule after the last iteration)."""
        msg = 'epoch:%d, iters:%d/%d ' % (epoch, iter, total_iter)
        for k, v in value1.items():
            msg += '%s: %.4f ' % (k, v)
        if value2:
            msg += '\n'
            for k, v in value2.items():
                msg += '%s:%.3f ' % (k, v)
        if iter == total_iter - 1:
            msg += '\n##########################################'
        print(msg)
        self.write_log(msg)
        return msg

    def write_log(self, msg):
        """utils/visualizer.py:82-87."""
        if self.log_path:
            os.makedirs(os.path.dirname(self.log_path) or '.', exist_ok=True)
            with open(self.log_path, 'a+') as log_file:
                log_file.write(msg + '\n')
