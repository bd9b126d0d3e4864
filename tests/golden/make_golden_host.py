def main():
    pass
