from mp3stego.decoder.decoder import Decoder
