"""utils/logger.py of the reference: the tab-separated training-summary file (plotting helpers left out)."""


class Logger(object):
    """Names header (tab-terminated fields) followed by one row of `{:.6f}` numbers per epoch (utils/logger.py:24-74)."""

    def __init__(self, fpath, title=None, resume=False):
        self.file = None
        self.resume = resume
        self.title = '' if title is None else title
        self.names, self.numbers = [], {}
        if fpath is not None:
            if resume:
                with open(fpath, 'r') as f:
                    self.names = f.readline().rstrip().split('\t')
                    self.numbers = {n: [] for n in self.names}
                    for line in f:
                        vals = line.rstrip().split('\t')
                        for i in range(len(vals)):
                            self.numbers[self.names[i]].append(vals[i])
                self.file = open(fpath, 'a')
            else:
                self.file = open(fpath, 'w')

    def set_names(self, names):
        self.numbers = {}
        self.names = list(names)
        for name in self.names:
            self.file.write(name)
            self.file.write('\t')
            self.numbers[name] = []
        self.file.write('\n')
        self.file.flush()

    def append(self, numbers):
        assert len(self.names) == len(numbers), 'Numbers do not match names'
        for index, num in enumerate(numbers):
            self.file.write("{0:.6f}".format(num))
            self.file.write('\t')
            self.numbers[self.names[index]].append(num)
        self.file.write('\n')
        self.file.flush()

    def close(self):
        if self.file is not None:
            self.file.close()
