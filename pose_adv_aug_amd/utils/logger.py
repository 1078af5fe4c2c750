"""Training-summary table of the reference's scripts (utils/logger.py:24-74, used by joint-train-pose-s-r-agent.py:131-133,
164-165,192): a text file whose first line holds the column names and every later line one epoch's numbers, fields
terminated by a tab, numbers printed with six decimals.  The matplotlib plotting half of the reference's class is outside
the hot path and not provided."""


class Logger(object):
    def __init__(self, fpath, title=None, resume=False):
        self.title = title or ''
        self.names = []
        self.numbers = {}
        self.file = None
        if fpath is None:
            return
        if resume:                                   # continue an existing table: read it back, then append
            with open(fpath) as existing:
                rows = [line.rstrip('\n').rstrip('\t').split('\t') for line in existing if line.strip()]
            if rows:
                self.names = rows[0]
                self.numbers = {name: [row[k] for row in rows[1:] if k < len(row)] for k, name in enumerate(self.names)}
        self.file = open(fpath, 'a' if resume else 'w')

    def _write_row(self, fields):
        self.file.write(''.join(f + '\t' for f in fields) + '\n')
        self.file.flush()

    def set_names(self, names):
        self.names = [str(n) for n in names]
        self.numbers = {n: [] for n in self.names}
        self._write_row(self.names)

    def append(self, numbers):
        if len(numbers) != len(self.names):
            raise ValueError('%d numbers for %d columns' % (len(numbers), len(self.names)))
        for name, value in zip(self.names, numbers):
            self.numbers[name].append(value)
        self._write_row(['%.6f' % float(v) for v in numbers])

    def close(self):
        if self.file is not None:
            self.file.close()
            self.file = None
