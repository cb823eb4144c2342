"""genie -- MI355X-native drop-in for the hot path of myscience/open-genie."""
