"""utils/visualizer.py of the reference, text part only (the visdom plots are outside the hot path): the per-iteration
log line and the log file."""
import os

RULE = '#' * 42


def format_log(epoch, it, total, meters, extra=None):
    """The reference's log line (utils/visualizer.py:66-78): 'epoch:E, iters:I/N k: v.vvvv ...', an optional second line
    'k:v.vvv ...', and a rule of 42 '#' after the last iteration of an epoch."""
    lines = ['epoch:%d, iters:%d/%d ' % (epoch, it, total) + ''.join('%s: %.4f ' % kv for kv in meters.items())]
    if extra:
        lines.append(''.join('%s:%.3f ' % kv for kv in extra.items()))
    if it == total - 1:
        lines.append(RULE)
    return '\n'.join(lines)


class Visualizer(object):
    def __init__(self, opt=None, log_path=None):
        self.opt = opt
        self.log_path = log_path

    def print_log(self, epoch, iter, total_iter, value1, value2=None):
        """utils/visualizer.py:66-80: console + log file."""
        msg = format_log(epoch, iter, total_iter, value1, value2)
        print(msg)
        self.write_log(msg)
        return msg

    def write_log(self, msg):
        """utils/visualizer.py:82-87."""
        if self.log_path:
            os.makedirs(os.path.dirname(self.log_path) or '.', exist_ok=True)
            with open(self.log_path, 'a+') as log_file:
                log_file.write(msg + '\n')
